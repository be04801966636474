"""PI2 / information-theoretic MPC -- reference PI2Optimizer (optimizers/pi2.py:9-11)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class PI2Optimizer(OptimizerBase):
    _engine_optimizer = L.OPT_PI2

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, max_iterations=5,
                 population_size=500, num_agents=5, lamda=1.0, **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=max_iterations,
                         num_agents=num_agents, env_action_space=env_action_space,
                         env_observation_space=env_observation_space, **engine_args)
        self._population_size = int(population_size)
        self._lamda = float(lamda)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size, lamda=self._lamda)
