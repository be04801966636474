"""Counterpart of the reference's SystemDynamicsHandler (dynamics_handlers/system_dynamics_handler.py).

Inference half (:7-161): which dynamics function, whether it is the true model, and the six normalisation
statistics; process_input / process_output are fused into the rollout kernels (GEMM-1 prologue / GEMM-3 epilogue).
Training half (:163-349): dataset assembly, train/validation split, freeze-after-first
normalisation, shuffled drop-remainder batches, MSE + Keras-Adam on the GPU (dynamics_functions/_train_torch.py);
the SavedModel checkpoint is replaced by `mlp.npz` + the reference's six `.npy` statistics files."""
import os

import numpy as np

_STATS = ("mean_states", "std_states", "mean_actions", "std_actions", "mean_targets", "std_targets")


class SystemDynamicsHandler:
    def __init__(self, env_action_space, env_observation_space, dynamics_function=None, true_model=False,
                 is_normalized=True, log_dir=None, tf_writer=None, save_model_frequency=1, saved_model_dir=None,
                 transform_targets_func=None, inverse_transform_targets_func=None):
        # custom target transforms (reference :128-161 applies inverse_transform_targets_func(states, raw)): honoured on the
        # torch-callable path, where process_output runs in torch (utils/device_functions.py); the built-in and HIP-source
        # models fuse the default delta transform and refuse a custom one (trajectory_evaluators/deterministic.py)
        self._transform_targets_func = transform_targets_func
        self._inverse_transform_targets_func = inverse_transform_targets_func
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
        # training state (:50-56)
        self._model_training_in = np.zeros((0, self._dim_U + self._dim_S), np.float32)
        self._model_validation_in = np.zeros((0, self._dim_U + self._dim_S), np.float32)
        self._model_training_out = np.zeros((0, self._dim_S), np.float32)
        self._model_validation_out = np.zeros((0, self._dim_S), np.float32)
        self._training_iter = 0
        self._refining_model_iter = 0
        self._first_time = True
        self.training_loss = None
        self.validation_loss = None
        if saved_model_dir is not None:
            self.load(saved_model_dir)
            self._first_time = False                     # :61 a loaded model keeps its statistics

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

    # -- process_input / process_output (system_dynamics_handler.py:97-161) as stand-alone calls: inside rollouts the same
    # arithmetic is fused into the kernels; called directly they run the device code through the C ABI
    def _io_engine(self):
        eng = self.__dict__.get("_io_eng")
        if eng is None:
            from .. import _lib as L
            from ..engine import Engine
            eng = self._io_eng = Engine(L.OPT_NONE, L.DYN_USER, L.REW_USER, self._env_action_space.low,
                                        self._env_action_space.high, dim_s=self._dim_S, num_agents=1, planning_horizon=1)
        return eng

    def process_input(self, states, actions):
        """states [B,S], actions [B,U] -> [B,S+U]: concat (true / un-normalised model) or z-scored concat."""
        return self._io_engine().process_input(states, actions, self.normalization_stats())

    def process_output(self, inputs_states, raw_output):
        """inputs_states [B,S], raw_output [B,S] -> absolute next states (de-normalise, then state + delta)."""
        return self._io_engine().process_output(inputs_states, raw_output, self.normalization_stats())

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

    # -- training (system_dynamics_handler.py:163-349) ---------------------------------------------------------------
    def _append_to_training_dataset(self, observations_trajectories, actions_trajectories, rewards_trajectories,
                                    validation_split=0.2, split_mask=None):
        """:292-331.  Episodes: observations [T+1, A, S], actions [T, A, U]; rows are (s_t, a_t) -> s_{t+1} - s_t,
        episode-major, then agent, then t.  Each row goes to the training set with probability 1 - validation_split
        (np.random.choice, :311-313); `split_mask` injects that draw (True = training row)."""
        acs_all = np.array(actions_trajectories)
        num_agents = acs_all.shape[2]
        d_in, d_out = [], []
        for obs, acs in zip(observations_trajectories, acs_all):
            obs = np.asarray(obs)
            for agent in range(num_agents):
                states = obs[:-1, agent]
                d_in.append(np.concatenate([states, acs[:, agent]], axis=-1))
                d_out.append(obs[1:, agent] - states)                      # default_transform_targets
        d_in = np.array(d_in, dtype=np.float32).reshape(-1, self._dim_U + self._dim_S)
        d_out = np.array(d_out, dtype=np.float32).reshape(-1, self._dim_S)
        if split_mask is None:
            split_mask = np.random.choice([False, True], size=d_in.shape[0], p=[validation_split, 1.0 - validation_split])
        split_mask = np.asarray(split_mask, bool)
        if split_mask.shape[0] != d_in.shape[0]:
            raise ValueError("split_mask needs one entry per transition (%d)" % d_in.shape[0])
        self._model_training_in = np.concatenate([self._model_training_in, d_in[split_mask]], axis=0)
        self._model_training_out = np.concatenate([self._model_training_out, d_out[split_mask]], axis=0)
        self._model_validation_in = np.concatenate([self._model_validation_in, d_in[~split_mask]], axis=0)
        self._model_validation_out = np.concatenate([self._model_validation_out, d_out[~split_mask]], axis=0)

    def _recompute_normalization(self):
        """:340-349 -- statistics of the TRAINING rows (population std)."""
        S = self._dim_S
        tin, tout = self._model_training_in, self._model_training_out
        self.set_normalization_stats(np.mean(tin[:, :S], axis=0), np.std(tin[:, :S], axis=0),
                                     np.mean(tin[:, S:], axis=0), np.std(tin[:, S:], axis=0),
                                     np.mean(tout, axis=0), np.std(tout, axis=0))

    def _normalize_data(self, data_in, data_out):
        """:333-338; un-normalised handlers train on the raw rows."""
        if not self._is_normalized:
            return data_in, data_out
        ms, ss, ma, sa, mt, st = self._stats
        S = self._dim_S
        s = (data_in[:, :S] - ms) / (ss + 1e-7)
        a = (data_in[:, S:] - ma) / (sa + 1e-7)
        t = (data_out - mt) / (st + 1e-7)
        return np.concatenate([s, a], axis=1).astype(np.float32), t.astype(np.float32)

    def train(self, observations_trajectories, actions_trajectories, rewards_trajectories, validation_split=0.2,
              batch_size=128, learning_rate=1e-3, epochs=30, nn_optimizer=None, *, device=None, seed=None,
              split_mask=None, permutations=None):
        """Reference signature (:163-166); `nn_optimizer`: None / "Adam" / "SGD" / "RMSprop" or a class of that name
        (the reference's callers only ever pass tf.keras.optimizers.Adam).  Keyword-only extras: `device` (default: the GPU -- training on the
        host has to be asked for explicitly with device="cpu"), and the injected random draws `split_mask`,
        `permutations` (one per epoch) / `seed` for reproducible runs."""
        if self._transform_targets_func is not None:
            raise NotImplementedError("training with a custom transform_targets_func is not built (default: next - state)")
        if self._is_true_model:
            raise Exception("the true model has nothing to train")
        # the reference instantiates `nn_optimizer(learning_rate=learning_rate)` (:261): a Keras optimizer CLASS (or its
        # name); Adam / SGD / RMSprop with their TF-2.0 defaults are built
        rule = "adam" if nn_optimizer is None else getattr(nn_optimizer, "__name__", str(nn_optimizer)).lower()
        if rule not in ("adam", "sgd", "rmsprop"):
            raise NotImplementedError("nn_optimizer %r: Adam, SGD and RMSprop (Keras defaults) are built" % (nn_optimizer,))
        fn = self._dynamics_function
        if fn is None or not hasattr(fn, "weights"):
            raise Exception("train() needs a DeterministicMLP dynamics function")
        import torch
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("SystemDynamicsHandler.train runs on the GPU and none is visible "
                                   "(pass device='cpu' explicitly to train on the host)")
            device = "cuda"
        self._append_to_training_dataset(observations_trajectories, actions_trajectories, rewards_trajectories,
                                         validation_split=validation_split, split_mask=split_mask)
        if self._first_time:                                               # :193-198 statistics are frozen after the first call
            if self._is_normalized:
                self._recompute_normalization()
            self._first_time = False
        tin, tout = self._normalize_data(self._model_training_in, self._model_training_out)
        vin, vout = self._normalize_data(self._model_validation_in, self._model_validation_out)
        from ..dynamics_functions._train_torch import DenseTrainer
        trainer = DenseTrainer(fn.weights, fn.biases, fn.activation_codes, device, learning_rate=learning_rate, rule=rule)  # fresh optimizer per call (:258)
        self.training_loss, self.validation_loss = trainer.fit(tin, tout, vin, vout, epochs, batch_size,
                                                               permutations=permutations, generator_seed=seed)
        fn.set_weights(*trainer.numpy_params())                            # bumps the version: evaluators re-upload
        self._version += 1
        self._refining_model_iter += 1                                     # :290
        self._training_iter += 1
        if self._training_iter % self._save_model_frequency == 0 and self._log_dir is not None:   # :212-241
            self.save(os.path.join(self._log_dir, "saved_model_%d" % self._refining_model_iter))
        return
