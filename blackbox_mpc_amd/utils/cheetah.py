"""HalfCheetah (modified obs, S=20) reward plugin -- reference tutorials/mujoco/cost_func.py:5-22."""
from .. import _lib as L


def reward_function(current_state, actions, next_state):
    raise NotImplementedError("cheetah reward_function is a device functor tag; it runs inside the engine "
                              "(DeterministicTrajectoryEvaluator.evaluate_next_reward)")


reward_function._bbmpc_reward_kind = L.REW_CHEETAH
cheetah_reward_function = reward_function
