"""Particle swarm -- reference PSOOptimizer (optimizers/pso.py:7-11)."""
from .. import _lib as L
from .optimizer_base import OptimizerBase


class PSOOptimizer(OptimizerBase):
    _engine_optimizer = L.OPT_PSO

    def __init__(self, env_action_space, env_observation_space, planning_horizon=50, max_iterations=5,
                 population_size=500, num_agents=5, c1=0.3, c2=0.5, w=0.2, initial_velocity_fraction=0.01,
                 **engine_args):
        super().__init__(name=None, planning_horizon=planning_horizon, max_iterations=max_iterations,
                         num_agents=num_agents, env_action_space=env_action_space,
                         env_observation_space=env_observation_space, **engine_args)
        self._population_size = int(population_size)
        self._c1, self._c2, self._w = float(c1), float(c2), float(w)
        self._initial_velocity_fraction = float(initial_velocity_fraction)

    def _engine_kwargs(self):
        return dict(population_size=self._population_size, pso_c1=self._c1, pso_c2=self._c2, pso_w=self._w,
                    pso_v0_fraction=self._initial_velocity_fraction)
