"""Pendulum-v0 analytic model + reward plugins (reference utils/pendulum.py).

In the engine these are device functors fused into the rollout kernels
(csrc/models.hpp); the Python objects are tags the host layer maps to
BBMPC_DYN_PENDULUM / BBMPC_REW_PENDULUM.  Calling them directly runs the same
device code through the C ABI."""
import numpy as np

from .. import _lib as L

_engines = {}


def _engine(quirks):
    from ..engine import Engine
    if quirks not in _engines:
        _engines[quirks] = Engine(L.OPT_NONE, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1,
                                  planning_horizon=1, quirks=quirks)
    return _engines[quirks]


def pendulum_reward_function(current_state, next_state, actions):
    """reference utils/pendulum.py:10-35, DECLARED argument order (current_state, next_state, actions).
    The evaluator calls its reward positionally as (cur, actions, next) (deterministic.py:65-66), which
    for this function means the action-cost term sees next_state (quirk Q1); the engine reproduces that
    inside rollouts.  A direct call like this one gets the declared semantics."""
    return _engine(L.FIX_Q1_REWARD_ARG_ORDER).evaluate_next_reward(current_state, next_state, actions)


pendulum_reward_function._bbmpc_reward_kind = L.REW_PENDULUM


class PendulumTrueModel:
    """reference utils/pendulum.py:38-92: x = [cos th, sin th, thdot, u] -> delta of the state."""
    _bbmpc_dynamics_kind = L.DYN_PENDULUM

    def __init__(self, name=None):
        self.name = name

    def __call__(self, x, train=False):
        x = np.asarray(x, dtype=np.float32)
        s, a = x[:, :3], x[:, 3:4]
        return _engine(0).predict_next_state(s, a) - s
