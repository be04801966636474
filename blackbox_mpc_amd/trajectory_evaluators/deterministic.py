"""DeterministicTrajectoryEvaluator -- same interface as the reference's
(trajectory_evaluators/deterministic.py:5-127), backed by the HIP rollout kernels."""
import numpy as np

from .. import _lib as L
from ..engine import Engine
from .evaluator_base import EvaluatorBase


def plugin_kinds(reward_function, handler):
    """Map the user-supplied callables to device functors (fused kernels need
    device code, arbitrary Python callables cannot run there)."""
    rk = getattr(reward_function, "_bbmpc_reward_kind", None)
    if rk is None:
        raise NotImplementedError(
            "reward_function %r has no device functor. Built-ins: blackbox_mpc_amd.utils.pendulum."
            "pendulum_reward_function, blackbox_mpc_amd.utils.cheetah.reward_function" % (reward_function,))
    dyn = handler._dynamics_function
    dk = getattr(dyn, "_bbmpc_dynamics_kind", None)
    if dk is None:
        raise NotImplementedError(
            "dynamics_function %r has no device functor. Built-ins: PendulumTrueModel, DeterministicMLP" % (dyn,))
    if dk == L.DYN_PENDULUM and not handler._is_true_model:
        raise Exception("PendulumTrueModel must be used with true_model=True")
    return dk, rk


def configure_dynamics(engine, handler):
    """Upload MLP weights + normalisation statistics when the dynamics are learned."""
    dyn = handler._dynamics_function
    if getattr(dyn, "_bbmpc_dynamics_kind", None) == L.DYN_MLP:
        engine.set_mlp(dyn.weights, dyn.biases, dyn.activation_codes, handler.normalization_stats())
    engine._dyn_version = (getattr(dyn, "_version", 0), handler._version)


def dynamics_stale(engine, handler):
    dyn = handler._dynamics_function
    return getattr(engine, "_dyn_version", None) != (getattr(dyn, "_version", 0), handler._version)


class DeterministicTrajectoryEvaluator(EvaluatorBase):
    def __init__(self, reward_function, system_dynamics_handler, quirks=0):
        super().__init__(reward_function=reward_function, system_dynamics_handler=system_dynamics_handler, name=None)
        self._quirks = int(quirks)
        self._engines = {}

    def _engine(self, num_agents, horizon):
        h = self._system_dynamics_handler
        key = (int(num_agents), int(horizon))
        eng = self._engines.get(key)
        if eng is None:
            dk, rk = plugin_kinds(self._reward_function, h)
            space = h._env_action_space
            eng = Engine(L.OPT_NONE, dk, rk, space.low, space.high, dim_s=h._dim_S, num_agents=key[0],
                         planning_horizon=key[1], quirks=self._quirks)
            self._engines[key] = eng
        if dynamics_stale(eng, h):
            configure_dynamics(eng, h)
        return eng

    def __call__(self, current_states, action_sequences, time_step=0):
        """current_states [A,S], action_sequences [N,A,H,U] -> rewards [N,A] (NaN -> -1e6)."""
        seq = np.asarray(action_sequences, np.float32)
        if seq.ndim != 4:
            raise ValueError("action_sequences must be [population, num_agents, planning_horizon, dim_U]")
        return self._engine(seq.shape[1], seq.shape[2]).evaluate(current_states, seq)

    def predict_next_state(self, current_states, current_actions):
        s = np.asarray(current_states, np.float32)
        return self._engine(1, 1).predict_next_state(s, np.asarray(current_actions, np.float32))

    def evaluate_next_reward(self, current_states, next_states, current_actions):
        return self._engine(1, 1).evaluate_next_reward(current_states, next_states, current_actions)
