"""Inference half of the reference's SystemDynamicsHandler
(dynamics_handlers/system_dynamics_handler.py:7-161): which dynamics function, whether it is the
true model, and the six normalisation statistics.  process_input / process_output are fused into
the rollout kernels (GEMM-1 prologue / GEMM-3 epilogue); training + SavedModel I/O are out of scope."""
import os

import numpy as np

_STATS = ("mean_states", "std_states", "mean_actions", "std_actions", "mean_targets", "std_targets")


class SystemDynamicsHandler:
    def __init__(self, env_action_space, env_observation_space, dynamics_function=None, true_model=False,
                 is_normalized=True, log_dir=None, tf_writer=None, save_model_frequency=1, saved_model_dir=None,
                 transform_targets_func=None, inverse_transform_targets_func=None):
        if transform_targets_func is not None or inverse_transform_targets_func is not None:
            raise NotImplementedError("only the default delta target transform (next = state + delta) is built")
        self._is_true_model = bool(true_model)
        self._dim_S = int(env_observation_space.shape[0])
        self._dim_U = int(env_action_space.shape[0])
        self._env_action_space = env_action_space
        self._env_observation_space = env_observation_space
        self._dynamics_function = dynamics_function
        self._is_normalized = bool(is_normalized)
        self._log_dir, self._tf_writer = log_dir, tf_writer
        self._save_model_frequency, self._saved_model_dir = save_model_frequency, saved_model_dir
        self._stats = None
        self._version = 0
        if saved_model_dir is not None:
            self.load(saved_model_dir)

    # -- normalisation statistics (system_dynamics_handler.py:84-95, 340-349) -----------------------
    def set_normalization_stats(self, mean_states, std_states, mean_actions, std_actions, mean_targets, std_targets):
        vals = [np.asarray(v, np.float32).reshape(-1) for v in
                (mean_states, std_states, mean_actions, std_actions, mean_targets, std_targets)]
        want = [self._dim_S, self._dim_S, self._dim_U, self._dim_U, self._dim_S, self._dim_S]
        for v, w, n in zip(vals, want, _STATS):
            if v.shape[0] != w:
                raise ValueError("%s must have %d entries" % (n, w))
        self._stats = vals
        self._version += 1

    def normalization_stats(self):
        if self._is_true_model or not self._is_normalized:
            return None
        if self._stats is None:
            raise Exception("dynamics handler is normalised but has no statistics yet "
                            "(set_normalization_stats / load)")
        return self._stats

    def load(self, saved_model_dir):
        """Counterpart of the reference's checkpoint load (:78-95): the six `.npy` statistics are read as
        written by the reference; the SavedModel graph is replaced by `mlp.npz` (DeterministicMLP.save)."""
        from ..dynamics_functions.deterministic_mlp import DeterministicMLP
        mlp = os.path.join(saved_model_dir, "mlp.npz")
        if os.path.exists(mlp):
            self._dynamics_function = DeterministicMLP.load(mlp)
        if self._is_normalized and all(os.path.exists(os.path.join(saved_model_dir, n + ".npy")) for n in _STATS):
            self.set_normalization_stats(*[np.load(os.path.join(saved_model_dir, n + ".npy")) for n in _STATS])

    def save(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        if hasattr(self._dynamics_function, "save"):
            self._dynamics_function.save(os.path.join(log_dir, "mlp.npz"))
        if self._stats is not None:
            for n, v in zip(_STATS, self._stats):
                np.save(os.path.join(log_dir, n + ".npy"), v)

    def train(self, *args, **kwargs):
        raise NotImplementedError("dynamics-model training is outside the rollout engine (SURVEY.md 8 f-3)")
