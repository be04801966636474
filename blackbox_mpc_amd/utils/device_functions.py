"""User-supplied reward / dynamics functions as device code.

The reference accepts any Python (TensorFlow) callable as `reward_function` / `dynamics_function`
(trajectory_evaluators/deterministic.py:13-18).  Here the rollouts run inside GPU kernels, so a user function is HIP
source compiled at run time (hiprtc) and called per row from the engine's step-wise evaluator:

    reward = HipRewardFunction('''
        __device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
            return -(cur[0] * cur[0]) - 0.1f * act[0] * act[0];
        }''')
    dynamics = HipDynamicsFunction('''
        __device__ void bbmpc_user_dynamics(const float* x, float* delta, int S, int U) {   // x = [state | action]
            delta[0] = 0.05f * x[1];  delta[1] = 0.05f * x[2];                              // returns next - state
        }''')
    policy = MPCPolicy(reward_function=reward, dynamics_function=dynamics, true_model=True, ...)

Argument order of the reward is the order of the reference's CALL, `reward_function(current_state, actions,
next_state)` (deterministic.py:65-66).  A user dynamics function is a true model: it returns the state delta and the
handler adds the state back (utils/transforms.py:34).  Both objects are also directly callable with NumPy batches
(they run the same device code through the C ABI), like the built-in plug-ins."""
import numpy as np

from .. import _lib as L


def check_source(kind, hip_source, dim_s, dim_u):
    """Compile only (no GPU needed); raises BBMPCError with the compiler log on failure."""
    L.check(L.lib.bbmpc_check_user_source(int(kind), hip_source.encode(), int(dim_s), int(dim_u)))


class _HipFunction:
    def __init__(self, hip_source):
        if not isinstance(hip_source, str) or not hip_source.strip():
            raise ValueError("hip_source must be a non-empty string of HIP device code")
        self.hip_source = hip_source
        self._engines = {}

    def _engine(self, dim_s, dim_u, low=None, high=None):
        from ..engine import Engine
        key = (int(dim_s), int(dim_u))
        eng = self._engines.get(key)
        if eng is None:
            lo = np.full(key[1], -1.0, np.float32) if low is None else low
            hi = np.full(key[1], 1.0, np.float32) if high is None else high
            eng = self._make(Engine, key[0], lo, hi)
            self._engines[key] = eng
        return eng


class HipRewardFunction(_HipFunction):
    """reward_function given as HIP source (`bbmpc_user_reward`)."""
    _bbmpc_reward_kind = L.REW_USER

    def _make(self, Engine, dim_s, lo, hi):
        # the dynamics kind is irrelevant for evaluate_next_reward; a user-dynamics handle needs no weights
        eng = Engine(L.OPT_NONE, L.DYN_USER, L.REW_USER, lo, hi, dim_s=dim_s, num_agents=1, planning_horizon=1)
        eng.set_reward_source(self.hip_source)
        return eng

    def __call__(self, current_state, actions, next_state):
        cur, act, nxt = (np.asarray(v, np.float32) for v in (current_state, actions, next_state))
        return self._engine(cur.shape[1], act.shape[1]).evaluate_next_reward(cur, nxt, act)


class HipDynamicsFunction(_HipFunction):
    """dynamics_function given as HIP source (`bbmpc_user_dynamics`); f(x[B,S+U], train) -> delta[B,S]."""
    _bbmpc_dynamics_kind = L.DYN_USER

    def __init__(self, hip_source, dim_s=None, dim_u=None):
        super().__init__(hip_source)
        self._dims = (dim_s, dim_u)

    def _make(self, Engine, dim_s, lo, hi):
        eng = Engine(L.OPT_NONE, L.DYN_USER, L.REW_USER, lo, hi, dim_s=dim_s, num_agents=1, planning_horizon=1)
        eng.set_dynamics_source(self.hip_source)
        return eng

    def __call__(self, x, train=False):
        x = np.asarray(x, np.float32)
        dim_s, dim_u = self._dims
        if dim_s is None or dim_u is None:
            raise ValueError("calling a HipDynamicsFunction directly needs dim_s / dim_u (constructor) to split x")
        s, a = x[:, :dim_s], x[:, dim_s:dim_s + dim_u]
        return self._engine(dim_s, dim_u).predict_next_state(s, a) - s
