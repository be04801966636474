"""Episode collection -- counterpart of the reference's utils/rollouts.py:10-139.

`perform_rollouts(env, number_of_rollouts, task_horizon, policy, exploration_noise)` keeps the reference's loop
(policy.reset(); env.reset(); for t: policy.act -> env.step) and return values (lists of per-episode observation /
action / reward arrays) and logs the mean action-selection time, the quantity the headline metric is defined on
(rollouts.py:92-101,133).  `env` is any object with the vector-env interface `reset() -> obs[A,S]`,
`step(actions[A,U]) -> (obs, reward, done, info)`; `ModelEnvironment` is such an environment whose dynamics is
the engine's own model on the GPU (gym is not a dependency).  `rollout_on_device` runs the whole closed loop inside
the engine (no per-step host round trip)."""
import logging
import time

import numpy as np


class ModelEnvironment:
    """Vector environment whose transition/reward are a DeterministicTrajectoryEvaluator's model (GPU)."""

    def __init__(self, trajectory_evaluator, start_states):
        self._ev = trajectory_evaluator
        self._start = np.asarray(start_states, np.float32)
        self._state = self._start.copy()
        h = getattr(trajectory_evaluator, "_system_dynamics_handler", None)
        self.action_space = getattr(h, "_env_action_space", None)
        self.observation_space = getattr(h, "_env_observation_space", None)

    def reset(self):
        self._state = self._start.copy()
        return self._state.copy()

    def step(self, actions):
        actions = np.asarray(actions, np.float32).reshape(self._state.shape[0], -1)
        nxt = self._ev.predict_next_state(self._state, actions)
        rew = self._ev.evaluate_next_reward(self._state, nxt, actions)
        self._state = nxt
        return nxt.copy(), rew, False, {}


def _sample(env, horizon, policy, episode_step, exploration_noise=False, tf_writer=None):
    from ..policies.model_free_base_policy import ModelFreeBasePolicy
    policy.reset()
    observations, actions, rewards, times, reward_sum = [env.reset()], [], [], [], 0
    for t in range(horizon):
        start = time.time()
        if isinstance(policy, ModelFreeBasePolicy):              # rollouts.py:93-101
            action = np.asarray(policy.act(observations[t], t))
        else:
            action, expected_obs, expected_reward = policy.act(observations[t], t, exploration_noise)
        times.append(time.time() - start)
        actions.append(action)
        obs, reward, done, info = env.step(action)
        observations.append(obs)
        rewards.append(reward)
        reward_sum += reward
    logging.info("Average action selection time: " + str(np.mean(times)))
    logging.info("Rollout length: " + str(len(actions)))
    return {"observations": np.array(observations), "actions": np.array(actions), "rewards": np.array(rewards),
            "reward_sum": reward_sum, "mean_act_time": float(np.mean(times))}


def perform_rollouts(env, number_of_rollouts, task_horizon, policy, exploration_noise=False, tf_writer=None,
                     start_episode=0):
    traj_obs, traj_acs, traj_rews = [], [], []
    for i in range(number_of_rollouts):
        s = _sample(env, task_horizon, policy, start_episode + i, exploration_noise, tf_writer)
        traj_obs.append(s["observations"])
        traj_acs.append(s["actions"])
        traj_rews.append(s["rewards"])
    return traj_obs, traj_acs, traj_rews


def rollout_on_device(policy, start_states, task_horizon, exploration_noise=False):
    """Whole episode inside the engine: returns (actions[T,A,U], observations[T+1,A,S], predicted_rewards[T,A])."""
    eng = policy._optimizer._require_engine()
    policy.reset()
    start_states = np.asarray(start_states, np.float32)
    a, n, r = eng.rollout_episode(start_states, task_horizon, exploration_noise)
    return a, np.concatenate([start_states[None], n], axis=0), r
