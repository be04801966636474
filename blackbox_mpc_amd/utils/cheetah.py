"""HalfCheetah (modified obs, S=20) reward plugin -- reference tutorials/mujoco/cost_func.py:5-22.

Inside rollouts it is a device functor fused into the MFMA kernels (csrc/models.hpp reward_generic, and the in-place
form in csrc/kernels_mlp.hpp); the Python object is the tag the host layer maps to BBMPC_REW_CHEETAH.  Calling it
directly runs the same device code through the C ABI (bbmpc_evaluate_next_reward), like the pendulum pair."""
import numpy as np

from .. import _lib as L

_engines = {}


def _engine(dim_s, dim_u):
    from ..engine import Engine
    key = (int(dim_s), int(dim_u))
    if key not in _engines:
        # evaluate_next_reward needs no model weights: a learned-dynamics handle without set_mlp is enough
        _engines[key] = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * key[1], [1.0] * key[1], dim_s=key[0],
                               num_agents=1, planning_horizon=1)
    return _engines[key]


def reward_function(current_state, actions, next_state):
    """r = -10*[s5 >= 0.2] - 10*[s6 >= 0] - 10*[s7 >= 0] + (s'17 - s17)/0.01 - 0*sum(a^2)   (cost_func.py:5-22)"""
    cur, act, nxt = (np.asarray(v, np.float32) for v in (current_state, actions, next_state))
    return _engine(cur.shape[1], act.shape[1]).evaluate_next_reward(cur, nxt, act)


reward_function._bbmpc_reward_kind = L.REW_CHEETAH
cheetah_reward_function = reward_function
