"""Cross-Entropy Method -- kwargs/defaults of the reference's CEMOptimizer (optimizers/cem.py:7-10)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class CEMOptimizer(OptimizerBase):
    _engine_optimizer = L.OPT_CEM

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, max_iterations=5,
                 population_size=500, num_elite=50, num_agents=5, epsilon=0.001, alpha=0.25, **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=max_iterations,
                         num_agents=num_agents, env_action_space=env_action_space,
                         env_observation_space=env_observation_space, **engine_args)
        self._population_size = int(population_size)
        self._num_elite = int(num_elite)
        self._epsilon = epsilon  # stored and never used, as in the reference (cem.py:53)
        self._alpha = float(alpha)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size, num_elite=self._num_elite, alpha=self._alpha)
