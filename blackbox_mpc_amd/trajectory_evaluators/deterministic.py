"""DeterministicTrajectoryEvaluator -- same interface as the reference's
(trajectory_evaluators/deterministic.py:5-127), backed by the HIP rollout kernels."""
import numpy as np

from .. import _lib as L
from ..engine import Engine
from .evaluator_base import EvaluatorBase


_REWARD_HELP = (
    "reward_function %r cannot run on the GPU: rollouts never leave it, so a custom reward is either device code -- HIP "
    "source wrapped in blackbox_mpc_amd.utils.device_functions.HipRewardFunction (or any object with a `hip_source` "
    "attribute) -- or a callable that works on PyTorch CUDA tensors, reward(current_state[B,S], actions[B,U], "
    "next_state[B,S]) -> [B] (%s).  Built-ins: blackbox_mpc_amd.utils.pendulum.pendulum_reward_function, "
    "blackbox_mpc_amd.utils.cheetah.reward_function")
_DYNAMICS_HELP = (
    "dynamics_function %r cannot run on the GPU: a custom model is either device code -- HIP source wrapped in "
    "blackbox_mpc_amd.utils.device_functions.HipDynamicsFunction -- or a callable / torch.nn.Module that works on "
    "PyTorch CUDA tensors, f(x[B,S+U]) -> [B,S] (%s).  Built-ins: PendulumTrueModel, DeterministicMLP")


def reward_plugin(reward_function, handler):
    """The object the engine is configured from: the function itself when it is a built-in or carries HIP source, else
    (a plain callable, what the reference's users pass, deterministic.py:13-18) its torch adapter -- probed once on a
    two-row batch of CUDA tensors, so that a host-only function is refused here and not in the middle of a control step."""
    from ..utils import device_functions as DF
    if getattr(reward_function, "_bbmpc_reward_kind", None) is not None or isinstance(getattr(reward_function, "hip_source", None), str):
        return reward_function
    if not callable(reward_function):
        raise NotImplementedError(_REWARD_HELP % (reward_function, "it is not callable"))
    dev = DF.gpu_for_callables()
    if dev is None:
        raise NotImplementedError(_REWARD_HELP % (reward_function, "no GPU / PyTorch here to run it on"))
    plug = DF.torch_plugin(reward_function, DF.TorchRewardFunction)
    try:
        plug.probe(handler._dim_S, handler._dim_U, dev)
    except Exception as ex:                               # noqa: BLE001
        raise NotImplementedError(_REWARD_HELP % (reward_function, "calling it with CUDA tensors failed: %s: %s"
                                                  % (type(ex).__name__, ex))) from ex
    return plug


def dynamics_plugin(handler):
    from ..utils import device_functions as DF
    dyn = handler._dynamics_function
    if getattr(dyn, "_bbmpc_dynamics_kind", None) is not None or isinstance(getattr(dyn, "hip_source", None), str):
        if getattr(handler, "_inverse_transform_targets_func", None) is not None:
            raise NotImplementedError("a custom inverse_transform_targets_func needs a dynamics_function that runs on PyTorch "
                                      "CUDA tensors (the built-in models and HIP-source models fuse the default delta transform)")
        return dyn
    if not callable(dyn):
        raise NotImplementedError(_DYNAMICS_HELP % (dyn, "it is not callable"))
    dev = DF.gpu_for_callables()
    if dev is None:
        raise NotImplementedError(_DYNAMICS_HELP % (dyn, "no GPU / PyTorch here to run it on"))
    plug = DF.torch_plugin(dyn, DF.TorchDynamicsFunction)
    try:
        plug.probe(handler, dev)
    except Exception as ex:                               # noqa: BLE001
        raise NotImplementedError(_DYNAMICS_HELP % (dyn, "calling it with CUDA tensors failed: %s: %s"
                                                    % (type(ex).__name__, ex))) from ex
    return plug


def plugin_kinds(reward_function, handler):
    """Map the user-supplied callables to the engine's plug-in kinds: built-in device functors, HIP
    source (BBMPC_*_USER, fused), or callables on torch CUDA tensors (BBMPC_*_USER with a device-memory callback)."""
    rp = reward_plugin(reward_function, handler)
    rk = getattr(rp, "_bbmpc_reward_kind", None)
    if rk is None:
        rk = L.REW_USER                       # any object that carries HIP source for bbmpc_user_reward
    dp = dynamics_plugin(handler)
    dk = getattr(dp, "_bbmpc_dynamics_kind", None)
    if dk is None:
        dk = L.DYN_USER
    hip_true_model = dk == L.DYN_PENDULUM or (dk == L.DYN_USER and isinstance(getattr(dp, "hip_source", None), str))
    if hip_true_model and not handler._is_true_model:
        raise Exception("%s must be used with true_model=True" % type(dp).__name__)
    return dk, rk


def configure_dynamics(engine, handler):
    """Upload MLP weights + normalisation statistics when the dynamics are learned; attach user dynamics."""
    dyn = dynamics_plugin(handler)
    if getattr(dyn, "_bbmpc_dynamics_kind", None) == L.DYN_MLP:
        engine.set_mlp(dyn.weights, dyn.biases, dyn.activation_codes, handler.normalization_stats())
    elif engine.cfg.dynamics == L.DYN_USER and isinstance(getattr(dyn, "hip_source", None), str):
        if getattr(engine, "_dyn_source", None) is not dyn.hip_source:
            engine.set_dynamics_source(dyn.hip_source)
            engine._dyn_source = dyn.hip_source
    elif engine.cfg.dynamics == L.DYN_USER:
        # a callable on torch CUDA tensors: rebuilt whenever the handler's statistics change (they are baked into it)
        engine.set_dynamics_callback(dyn.make_callback(handler, engine.device))
    engine._dyn_version = (getattr(handler._dynamics_function, "_version", 0), handler._version)


def configure_reward(engine, reward_function, handler=None):
    """Compile + attach the user's reward device function (or its torch callback), once per engine."""
    if engine.cfg.reward != L.REW_USER:
        return
    if isinstance(getattr(reward_function, "hip_source", None), str):
        if getattr(engine, "_rew_source", None) is not reward_function.hip_source:
            engine.set_reward_source(reward_function.hip_source)
            engine._rew_source = reward_function.hip_source
        return
    from ..utils import device_functions as DF
    plug = reward_function if isinstance(reward_function, DF.TorchRewardFunction) else DF.torch_plugin(reward_function, DF.TorchRewardFunction)
    if getattr(engine, "_rew_plugin", None) is not plug:
        engine.set_reward_callback(plug.make_callback(engine.S, engine.U, engine.device))
        engine._rew_plugin = plug


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
