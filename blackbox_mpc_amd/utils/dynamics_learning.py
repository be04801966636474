"""learn_dynamics_from_policy -- counterpart of the reference's utils/dynamics_learning.py:7-90: collect episodes
with a policy, then fit the dynamics model on them (SystemDynamicsHandler.train, on the GPU)."""
from ..dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler
from .rollouts import perform_rollouts


def learn_dynamics_from_policy(env, policy, number_of_rollouts, task_horizon, dynamics_function=None,
                               system_dynamics_handler=None, epochs=30, learning_rate=1e-3, validation_split=0.2,
                               batch_size=128, is_normalized=True, nn_optimizer=None, tf_writer=None,
                               exploration_noise=False, log_dir=None, save_model_frequency=1, saved_model_dir=None,
                               start_episode=0, **train_args):
    """Same arguments as the reference; `train_args` (device=, seed=, ...) are forwarded to `train`."""
    if system_dynamics_handler is None:
        system_dynamics_handler = SystemDynamicsHandler(env_action_space=env.action_space,
                                                        env_observation_space=env.observation_space,
                                                        true_model=False, dynamics_function=dynamics_function,
                                                        tf_writer=tf_writer, is_normalized=is_normalized,
                                                        log_dir=log_dir, save_model_frequency=save_model_frequency,
                                                        saved_model_dir=saved_model_dir)
    traj_obs, traj_acs, traj_rews = perform_rollouts(env, number_of_rollouts, task_horizon, policy,
                                                     exploration_noise=exploration_noise, tf_writer=tf_writer,
                                                     start_episode=start_episode)
    system_dynamics_handler.train(traj_obs, traj_acs, traj_rews, validation_split=validation_split,
                                  batch_size=batch_size, learning_rate=learning_rate, epochs=epochs,
                                  nn_optimizer=nn_optimizer, **train_args)
    return system_dynamics_handler
