"""reference policies/model_free_base_policy.py: base of policies that need no model (act(obs, t) -> action)."""


class ModelFreeBasePolicy:
    def act(self, observations, t, exploration_noise=False):
        raise Exception("act is not implemented")

    def reset(self):
        raise Exception("reset is not implemented")
