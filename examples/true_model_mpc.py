"""Counterpart of the reference's tutorials/true_model_mpc + model_based_RL/tutorial_one: MPC on the pendulum with the
known analytic model, for each of the six optimizers.  Only the imports differ from the reference script; gym is not
needed -- the "real system" is the same analytic pendulum stepped on the GPU (utils.rollouts.ModelEnvironment).

    python examples/true_model_mpc.py            # needs an MI355X and the built libbbmpc.so

CMA-ES runs with the reference's semantics (agents coupled through the summed rewards and one joint covariance, samples
z @ (B @ D): SURVEY quirks Q5/Q6), which is what the parity tests pin -- not a tuned swing-up controller.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackbox_mpc_amd import Box                                                     # noqa: E402
from blackbox_mpc_amd.dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler   # noqa: E402
from blackbox_mpc_amd.policies.mpc_policy import MPCPolicy                            # noqa: E402
from blackbox_mpc_amd.trajectory_evaluators.deterministic import DeterministicTrajectoryEvaluator  # noqa: E402
from blackbox_mpc_amd.utils.pendulum import PendulumTrueModel, pendulum_reward_function  # noqa: E402
from blackbox_mpc_amd.utils.rollouts import ModelEnvironment, perform_rollouts       # noqa: E402

action_space = Box(low=[-2.0], high=[2.0])
observation_space = Box(low=[-1.0, -1.0, -8.0], high=[1.0, 1.0, 8.0])
num_agents, task_horizon = 4, 200

# hanging down (theta = pi), at rest: the swing-up task
start = np.tile(np.array([[-1.0, 0.0, 0.0]], np.float32), (num_agents, 1))
handler = SystemDynamicsHandler(action_space, observation_space, dynamics_function=PendulumTrueModel(), true_model=True)
env = ModelEnvironment(DeterministicTrajectoryEvaluator(pendulum_reward_function, handler), start)

for name, kwargs in [("RandomSearch", dict(population_size=1024)),
                     ("CEM", dict(population_size=500, num_elite=50, max_iterations=5)),
                     ("PI2", dict(population_size=500, max_iterations=5)),
                     ("PSO", dict(population_size=500, max_iterations=5)),
                     ("SPSA", dict(population_size=500, max_iterations=5)),
                     ("CMA-ES", dict(population_size=500, num_elite=50, max_iterations=5))]:
    mpc_policy = MPCPolicy(reward_function=pendulum_reward_function, env_action_space=action_space,
                           env_observation_space=observation_space, true_model=True,
                           dynamics_function=PendulumTrueModel(), optimizer_name=name, num_agents=num_agents,
                           planning_horizon=30, **kwargs)
    mpc_policy.act(start, 0)      # first use of an optimizer loads its kernels (and, once per process, the sampling tables): not timed
    t0 = time.time()
    traj_obs, traj_acs, traj_rews = perform_rollouts(env, 1, task_horizon, mpc_policy)
    dt = time.time() - t0
    final_cos = traj_obs[0][-1][:, 0]
    print("%-12s episode reward %8.1f   upright at the end: %d/%d agents   %.0f control steps/s (host in / host out)"
          % (name, float(np.mean(np.sum(traj_rews[0], axis=0))), int(np.sum(final_cos > 0.95)), num_agents,
             task_horizon / dt))
