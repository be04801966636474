"""SPSA -- reference SPSAOptimizer (optimizers/spsa.py:7-12)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class SPSAOptimizer(OptimizerBase):
    _engine_optimizer = L.OPT_SPSA

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, max_iterations=5,
                 population_size=500, num_agents=5, alpha=0.602, gamma=0.101, a_par=0.01, noise_parameter=0.3,
                 **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=max_iterations,
                         num_agents=num_agents, env_action_space=env_action_space,
                         env_observation_space=env_observation_space, **engine_args)
        self._population_size = int(population_size)
        self._alpha, self._gamma = float(alpha), float(gamma)
        self._a_par, self._noise_parameter = float(a_par), float(noise_parameter)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size, spsa_alpha=self._alpha, spsa_gamma=self._gamma,
                    spsa_a=self._a_par, spsa_c=self._noise_parameter)
