"""Counterpart of the reference's tutorials/learn_dynamics and model_based_RL: learn an MLP dynamics model from random
rollouts, refine it with MPC rollouts (iterative MPC), then control the real system through the learned model.
Only the imports differ from the reference scripts (and `tf.math.tanh` becomes "tanh"); the "real system" is the
analytic pendulum stepped on the GPU.

    python examples/learn_dynamics_and_control.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbox_mpc_amd import Box                                                     # noqa: E402
from blackbox_mpc_amd.dynamics_functions.deterministic_mlp import DeterministicMLP   # noqa: E402
from blackbox_mpc_amd.dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler   # noqa: E402
from blackbox_mpc_amd.policies import RandomPolicy                                   # noqa: E402
from blackbox_mpc_amd.trajectory_evaluators.deterministic import DeterministicTrajectoryEvaluator  # noqa: E402
from blackbox_mpc_amd.utils.iterative_mpc import learn_dynamics_iteratively_w_mpc    # noqa: E402
from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function  # noqa: E402
from blackbox_mpc_amd.utils.rollouts import ModelEnvironment, perform_rollouts       # noqa: E402

action_space = Box(low=[-2.0], high=[2.0])
observation_space = Box(low=[-1.0, -1.0, -8.0], high=[1.0, 1.0, 8.0])
num_agents, task_horizon = 10, 200

rng = np.random.default_rng(0)
theta0 = rng.uniform(-np.pi, np.pi, num_agents)
start = np.stack([np.cos(theta0), np.sin(theta0), rng.uniform(-1, 1, num_agents)], axis=1).astype(np.float32)
true_handler = SystemDynamicsHandler(action_space, observation_space, dynamics_function=PendulumTrueModel(), true_model=True)
env = ModelEnvironment(DeterministicTrajectoryEvaluator(pendulum_reward_function, true_handler), start)

dynamics_function = DeterministicMLP(layers=[4, 32, 32, 32, 3], activation_functions=["tanh", "tanh", "tanh", None], seed=0)
handler, mpc_policy = learn_dynamics_iteratively_w_mpc(
    env, number_of_initial_rollouts=5, number_of_rollouts_for_refinement=2, number_of_refinement_steps=3,
    task_horizon=task_horizon, env_action_space=action_space, env_observation_space=observation_space,
    initial_policy=RandomPolicy(num_agents, action_space, seed=0), planning_horizon=30,
    reward_function=pendulum_reward_function, optimizer_name="CEM", num_agents=num_agents,
    dynamics_function=dynamics_function, epochs=30, learning_rate=1e-3, batch_size=128, train_args={"seed": 0},
    population_size=500, num_elite=50, max_iterations=5)
print("transitions collected: %d train + %d validation; last validation loss %.4f"
      % (handler._model_training_in.shape[0], handler._model_validation_in.shape[0], handler.validation_loss[-1]))

traj_obs, traj_acs, traj_rews = perform_rollouts(env, 1, task_horizon, mpc_policy)
print("MPC through the learned model on the real system: mean episode reward %.1f, upright at the end: %d/%d agents"
      % (float(np.mean(np.sum(traj_rews[0], axis=0))), int(np.sum(traj_obs[0][-1][:, 0] > 0.95)), num_agents))
