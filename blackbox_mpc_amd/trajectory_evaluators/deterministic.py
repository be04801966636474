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
    if rk is None and isinstance(getattr(reward_function, "hip_source", None), str):
        rk = L.REW_USER                       # any object that carries HIP source for bbmpc_user_reward
    if rk is None:
        raise NotImplementedError(
            "reward_function %r is a host callable: rollouts run inside GPU kernels, so a custom reward must be device "
            "code -- wrap HIP source in blackbox_mpc_amd.utils.device_functions.HipRewardFunction (or give the object a "
            "`hip_source` attribute).  Built-ins: blackbox_mpc_amd.utils.pendulum.pendulum_reward_function, "
            "blackbox_mpc_amd.utils.cheetah.reward_function" % (reward_function,))
    dyn = handler._dynamics_function
    dk = getattr(dyn, "_bbmpc_dynamics_kind", None)
    if dk is None and isinstance(getattr(dyn, "hip_source", None), str):
        dk = L.DYN_USER
    if dk is None:
        raise NotImplementedError(
            "dynamics_function %r is a host callable: a custom model must be device code -- wrap HIP source in "
            "blackbox_mpc_amd.utils.device_functions.HipDynamicsFunction.  Built-ins: PendulumTrueModel, "
            "DeterministicMLP" % (dyn,))
    if dk in (L.DYN_PENDULUM, L.DYN_USER) and not handler._is_true_model:
        raise Exception("%s must be used with true_model=True" % type(dyn).__name__)
    return dk, rk


def configure_dynamics(engine, handler):
    """Upload MLP weights + normalisation statistics when the dynamics are learned."""
    dyn = handler._dynamics_function
    if getattr(dyn, "_bbmpc_dynamics_kind", None) == L.DYN_MLP:
        engine.set_mlp(dyn.weights, dyn.biases, dyn.activation_codes, handler.normalization_stats())
    elif engine.cfg.dynamics == L.DYN_USER and getattr(engine, "_dyn_source", None) is not dyn.hip_source:
        engine.set_dynamics_source(dyn.hip_source)
        engine._dyn_source = dyn.hip_source
    engine._dyn_version = (getattr(dyn, "_version", 0), handler._version)


def configure_reward(engine, reward_function):
    """Compile + attach the user's reward device function, once per engine."""
    if engine.cfg.reward == L.REW_USER and getattr(engine, "_rew_source", None) is not reward_function.hip_source:
        engine.set_reward_source(reward_function.hip_source)
        engine._rew_source = reward_function.hip_source


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
            configure_reward(eng, self._reward_function)
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
