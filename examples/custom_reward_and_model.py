"""Custom reward / dynamics functions (the reference accepts any TF callable, trajectory_evaluators/deterministic.py:13-18).
Here they are HIP device code compiled at run time: a cart-like double integrator steered to the origin with CEM."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbox_mpc_amd.policies import MPCPolicy
from blackbox_mpc_amd.spaces import Box
from blackbox_mpc_amd.utils.device_functions import HipDynamicsFunction, HipRewardFunction

reward = HipRewardFunction("""
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    return -(nxt[0] * nxt[0] + 0.1f * nxt[1] * nxt[1]) - 0.01f * act[0] * act[0];     // (current_state, actions, next_state)
}""")
model = HipDynamicsFunction("""
__device__ void bbmpc_user_dynamics(const float* x, float* delta, int S, int U) {       // x = [pos, vel | force] -> next - state
    delta[0] = 0.05f * x[1];
    delta[1] = 0.05f * x[2];
}""", dim_s=2, dim_u=1)

policy = MPCPolicy(reward_function=reward, dynamics_function=model, true_model=True, env_action_space=Box([-1.0], [1.0]),
                   env_observation_space=Box([-10, -10], [10, 10]), optimizer_name="CEM", num_agents=1, planning_horizon=40,
                   population_size=512, max_iterations=4, num_elite=32)
obs = np.array([[2.0, 0.0]], np.float32)
for t in range(120):
    action, predicted_next, predicted_reward = policy.act(obs, t)
    obs = predicted_next                       # the model is the environment here
    if t % 20 == 0:
        print("t=%3d  pos % .3f  vel % .3f  force % .3f" % (t, obs[0, 0], obs[0, 1], action[0, 0]))
assert abs(obs[0, 0]) < 0.2
print("reached the origin")
