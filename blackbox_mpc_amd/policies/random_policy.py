"""reference policies/random_policy.py:5-53: uniform random actions, used to collect the first episodes the
dynamics model is trained on.  (The reference passes (high, low) as (minval, maxval) to tf.random.uniform, :15-18,
i.e. a = high + u * (low - high): still uniform over the action box; kept as is.)"""
import numpy as np

from .model_free_base_policy import ModelFreeBasePolicy


class RandomPolicy(ModelFreeBasePolicy):
    def __init__(self, number_of_agents, env_action_space, seed=None):
        self._num_of_agents = int(number_of_agents)
        self._action_lower_bound = np.asarray(env_action_space.high, np.float32)
        self._action_upper_bound = np.asarray(env_action_space.low, np.float32)
        self._rng = np.random.default_rng(seed)

    def act(self, observations, t, exploration_noise=False):
        u = self._rng.random((self._num_of_agents,) + self._action_lower_bound.shape, dtype=np.float32)
        return (self._action_lower_bound + u * (self._action_upper_bound - self._action_lower_bound)).astype(np.float32)

    def reset(self):
        return
