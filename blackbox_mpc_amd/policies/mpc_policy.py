"""MPCPolicy -- drop-in for the reference's policies/mpc_policy.py:8-245.

Same constructor kwargs, `act / reset / switch_optimizer`, same error behaviour; one `act` is one
C-ABI call (state up, all optimizer iterations on the GPU, packed action/next-state/reward down)."""
import numpy as np

from ..dynamics_handlers.system_dynamics_handler import SystemDynamicsHandler
from ..trajectory_evaluators.deterministic import DeterministicTrajectoryEvaluator
from .model_based_base_policy import ModelBasedBasePolicy


def _optimizer_class(optimizer_name):
    from .. import optimizers as O
    return {"CEM": O.CEMOptimizer, "CMA-ES": O.CMAESOptimizer, "PI2": O.PI2Optimizer, "PSO": O.PSOOptimizer,
            "SPSA": O.SPSAOptimizer, "RandomSearch": O.RandomSearchOptimizer}.get(optimizer_name)


class MPCPolicy(ModelBasedBasePolicy):
    def __init__(self, trajectory_evaluator=None, optimizer=None, tf_writer=None, log_dir=None,
                 reward_function=None, env_action_space=None, env_observation_space=None,
                 dynamics_function=None, dynamics_handler=None, true_model=False, optimizer_name=None,
                 num_agents=None, save_model_frequency=1, saved_model_dir=None, **optimizer_args):
        if trajectory_evaluator is None:
            if dynamics_handler is None:
                dynamics_handler = SystemDynamicsHandler(
                    env_action_space=env_action_space, env_observation_space=env_observation_space,
                    true_model=true_model, dynamics_function=dynamics_function, log_dir=log_dir,
                    tf_writer=tf_writer, save_model_frequency=save_model_frequency,
                    saved_model_dir=saved_model_dir)
            trajectory_evaluator = DeterministicTrajectoryEvaluator(reward_function=reward_function,
                                                                    system_dynamics_handler=dynamics_handler)
        super(MPCPolicy, self).__init__(trajectory_evaluator=trajectory_evaluator)
        if optimizer is None:
            if num_agents is None:
                raise Exception("Please Specify Num Of Agents in the MPC")
            cls = _optimizer_class(optimizer_name)
            if cls is not None:
                optimizer = cls(env_action_space=env_action_space, env_observation_space=env_observation_space,
                                num_agents=num_agents, **optimizer_args)
        self._optimizer = optimizer
        self._tf_writer = tf_writer
        self._trajectory_evaluator = trajectory_evaluator
        # an unknown optimizer_name leaves optimizer None -> AttributeError here, as in the reference (:120)
        self._optimizer.set_trajectory_evaluator(trajectory_evaluator)
        self._act_call_counter = 0

    def act(self, observations, t, exploration_noise=False):
        observations = np.asarray(observations)
        batched = observations
        if observations.ndim == 1:
            batched = np.tile(observations[None], (self._optimizer._num_agents, 1))
        action, next_observations, rewards = self._optimizer(batched, t, exploration_noise)
        self._act_call_counter += 1
        if observations.ndim == 1:
            return action[0], next_observations[0], rewards[0]
        return action, next_observations, rewards

    def reset(self):
        self._optimizer.reset()

    def switch_optimizer(self, optimizer=None, optimizer_name='', **optimizer_args):
        if optimizer is None:
            cls = _optimizer_class(optimizer_name)
            if cls is not None:
                old = self._optimizer
                self._optimizer = cls(env_action_space=old._env_action_space,
                                      env_observation_space=old._env_observation_space,
                                      num_agents=old._num_agents, **optimizer_args)
        else:
            self._optimizer = optimizer
        self._optimizer.set_trajectory_evaluator(self._trajectory_evaluator)
        return
