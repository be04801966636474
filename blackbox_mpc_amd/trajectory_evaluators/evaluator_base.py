"""reference trajectory_evaluators/evaluator_base.py:4-85."""


class EvaluatorBase:
    def __init__(self, reward_function, system_dynamics_handler, name=None):
        self.name = name
        self._reward_function = reward_function
        self._system_dynamics_handler = system_dynamics_handler

    def __call__(self, current_states, action_sequences, time_step):
        raise Exception("__call__ function is not implemented yet")

    def predict_next_state(self, current_state, current_action):
        raise Exception("predict_next_state function is not implemented yet")

    def evaluate_next_reward(self, current_state, next_state, current_action):
        raise Exception("evaluate_next_reward function is not implemented yet")
