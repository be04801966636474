"""OptimizerBase -- the reference's optimizer interface (optimizers/optimizer_base.py:5-115).

A subclass only declares which engine optimizer it is and its hyper-parameters; the whole
per-control-step loop (sample -> rollout -> reduce -> refit, every iteration, plus the
exploration-noise / predicted-next-state tail of __call__) runs on the GPU through the C ABI."""
import numpy as np

from .. import _lib as L
from ..engine import Engine


class OptimizerBase:
    _engine_optimizer = L.OPT_NONE

    def __init__(self, name, planning_horizon, max_iterations, num_agents, env_action_space,
                 env_observation_space, seed=0, quirks=0, agent_offset=0, num_agents_global=None, device=-1,
                 population_offset=0, population_global=0):
        self.name = name
        self._planning_horizon = int(planning_horizon)
        self._env_action_space = env_action_space
        self._env_observation_space = env_observation_space
        self._dim_U = int(env_action_space.shape[0])
        self._dim_S = int(env_observation_space.shape[0])
        self._action_upper_bound = np.asarray(env_action_space.high, np.float32)
        self._action_lower_bound = np.asarray(env_action_space.low, np.float32)
        self._num_agents = int(num_agents)
        self._max_iterations = max_iterations
        self._trajectory_evaluator = None
        self._exploration_variance = (np.square(self._action_lower_bound - self._action_upper_bound) / 16) * 0.05
        self._exploration_mean = (self._action_upper_bound + self._action_lower_bound) / 2
        self._seed, self._quirks = int(seed), int(quirks)
        self._agent_offset, self._num_agents_global, self._device = int(agent_offset), num_agents_global, int(device)
        self._population_offset, self._population_global = int(population_offset), int(population_global or 0)
        self._engine = None

    # hyper-parameters forwarded to bbmpc_config; overridden by subclasses
    def _engine_kwargs(self):
        return {}

    def _optimize(self, current_state, time_step):
        raise Exception("__call__ function is not implemented yet")

    def _require_engine(self):
        if self._engine is None:
            if type(self)._engine_optimizer == L.OPT_NONE:
                raise Exception("__call__ function is not implemented yet")
            raise Exception("trajectory evaluator is not set; call set_trajectory_evaluator first")
        eng = self._engine
        h = self._trajectory_evaluator._system_dynamics_handler
        # inline staleness test (a learned model that was refitted / reloaded is re-uploaded before the next step)
        if eng.__dict__.get("_dyn_version") != (getattr(h._dynamics_function, "_version", 0), h._version):
            from ..trajectory_evaluators.deterministic import configure_dynamics
            configure_dynamics(eng, h)
        return eng

    def __call__(self, current_state, time_step=0, add_exploration_noise=False):
        """(current_state[A,S], time_step, add_exploration_noise) ->
        (action[A,U], next_state[A,S], rewards_of_next_state[A])   optimizer_base.py:55-95"""
        return self._require_engine().optimize(current_state, time_step, add_exploration_noise)

    def reset(self):
        if type(self)._engine_optimizer == L.OPT_NONE:
            raise Exception("reset function is not implemented yet")
        if self._engine is not None:
            self._engine.reset()

    def set_trajectory_evaluator(self, trajectory_evaluator):
        from ..trajectory_evaluators.deterministic import configure_dynamics, configure_reward, plugin_kinds
        self._trajectory_evaluator = trajectory_evaluator
        if type(self)._engine_optimizer == L.OPT_NONE:
            return
        h = trajectory_evaluator._system_dynamics_handler
        dk, rk = plugin_kinds(trajectory_evaluator._reward_function, h)
        if self._engine is not None:
            self._engine.close()
        quirks = self._quirks | int(getattr(trajectory_evaluator, "_quirks", 0))
        self._engine = Engine(type(self)._engine_optimizer, dk, rk, self._action_lower_bound, self._action_upper_bound,
                              dim_s=self._dim_S, num_agents=self._num_agents, planning_horizon=self._planning_horizon,
                              max_iterations=self._max_iterations or 0, seed=self._seed, quirks=quirks,
                              agent_offset=self._agent_offset, num_agents_global=self._num_agents_global,
                              device=self._device, population_offset=self._population_offset,
                              population_global=self._population_global, **self._engine_kwargs())
        configure_dynamics(self._engine, h)
        configure_reward(self._engine, trajectory_evaluator._reward_function)
        return
