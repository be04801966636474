"""learn_dynamics_iteratively_w_mpc -- counterpart of the reference's utils/iterative_mpc.py:11-174: fit an initial
model from `initial_policy` episodes, then alternate (collect episodes with the MPC policy planning through the
current model) / (refit the model on everything collected so far).  Every refit bumps the model version, so the
policy's evaluator re-uploads the weights into the rollout engine before the next control step."""
import logging

from ..dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler
from ..policies.mpc_policy import MPCPolicy
from .dynamics_learning import learn_dynamics_from_policy


def learn_dynamics_iteratively_w_mpc(env, number_of_initial_rollouts, number_of_rollouts_for_refinement,
                                     number_of_refinement_steps, task_horizon, env_action_space=None,
                                     env_observation_space=None, initial_policy=None, refinement_policy=None,
                                     planning_horizon=None, reward_function=None, is_normalized=True,
                                     optimizer_name='CEM', optimizer=None, num_agents=None, nn_optimizer=None,
                                     dynamics_function=None, system_dynamics_handler=None, log_dir=None,
                                     tf_writer=None, save_model_frequency=1, saved_model_dir=None,
                                     exploration_noise=False, epochs=30, learning_rate=1e-3, validation_split=0.2,
                                     batch_size=128, start_episode=0, train_args=None, **optimizer_args):
    """Same arguments as the reference (+ `train_args`, a dict forwarded to SystemDynamicsHandler.train).
    Returns (system_dynamics_handler, refinement_policy)."""
    train_args = dict(train_args or {})
    if number_of_initial_rollouts > 0:
        system_dynamics_handler = learn_dynamics_from_policy(
            env=env, policy=initial_policy, number_of_rollouts=number_of_initial_rollouts,
            task_horizon=task_horizon, dynamics_function=dynamics_function,
            system_dynamics_handler=system_dynamics_handler, epochs=epochs, learning_rate=learning_rate,
            validation_split=validation_split, batch_size=batch_size, is_normalized=is_normalized,
            nn_optimizer=nn_optimizer, tf_writer=tf_writer, exploration_noise=exploration_noise, log_dir=log_dir,
            save_model_frequency=save_model_frequency, saved_model_dir=saved_model_dir, **train_args)
        logging.info("Trained initial system model")
    elif system_dynamics_handler is None:
        system_dynamics_handler = SystemDynamicsHandler(
            env_action_space=env_action_space, env_observation_space=env_observation_space, true_model=False,
            dynamics_function=dynamics_function, tf_writer=tf_writer, is_normalized=is_normalized, log_dir=log_dir,
            save_model_frequency=save_model_frequency, saved_model_dir=saved_model_dir)
    if refinement_policy is None:
        refinement_policy = MPCPolicy(reward_function=reward_function, env_action_space=env_action_space,
                                      env_observation_space=env_observation_space,
                                      dynamics_handler=system_dynamics_handler, optimizer=optimizer,
                                      optimizer_name=optimizer_name, num_agents=num_agents,
                                      planning_horizon=planning_horizon, tf_writer=tf_writer, **optimizer_args)
    for i in range(number_of_refinement_steps):
        system_dynamics_handler = learn_dynamics_from_policy(
            env=env, policy=refinement_policy, number_of_rollouts=number_of_rollouts_for_refinement,
            task_horizon=task_horizon, system_dynamics_handler=system_dynamics_handler, epochs=epochs,
            learning_rate=learning_rate, validation_split=validation_split, batch_size=batch_size,
            is_normalized=is_normalized, nn_optimizer=nn_optimizer, tf_writer=tf_writer,
            exploration_noise=exploration_noise,
            start_episode=start_episode + (number_of_rollouts_for_refinement * i), **train_args)
    return system_dynamics_handler, refinement_policy
