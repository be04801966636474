"""Random shooting -- reference RandomSearchOptimizer (optimizers/random_search.py:7-8)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class RandomSearchOptimizer(OptimizerBase):
    _engine_optimizer = L.OPT_RANDOM_SEARCH

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, population_size=1024,
                 num_agents=5, **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=None, num_agents=num_agents,
                         env_action_space=env_action_space, env_observation_space=env_observation_space,
                         **engine_args)
        self._population_size = int(population_size)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size)
