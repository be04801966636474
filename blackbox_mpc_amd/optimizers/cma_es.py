"""CMA-ES -- reference CMAESOptimizer (optimizers/cma_es.py:7-10)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class CMAESOptimizer(OptimizerBase):
    _engine_optimizer = L.OPT_CMAES

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, max_iterations=5,
                 population_size=500, num_elite=50, num_agents=5, alpha_cov=2.0, h_sigma=1.0, **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=max_iterations,
                         num_agents=num_agents, env_action_space=env_action_space,
                         env_observation_space=env_observation_space, **engine_args)
        self._population_size = int(population_size)
        self._num_elite = int(num_elite)
        self._alpha_cov, self._h_sigma = float(alpha_cov), float(h_sigma)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size, num_elite=self._num_elite,
                    cma_alpha_cov=self._alpha_cov, cma_h_sigma=self._h_sigma)
