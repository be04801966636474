"""reference policies/model_based_base_policy.py:1-48."""


class ModelBasedBasePolicy(object):
    def __init__(self, trajectory_evaluator):
        self._trajectory_evaluator = trajectory_evaluator

    def act(self, observations, t, exploration_noise=False):
        raise Exception("act function is not implemented yet")

    def reset(self):
        raise Exception("reset function is not implemented yet")
