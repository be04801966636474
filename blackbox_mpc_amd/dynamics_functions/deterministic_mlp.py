"""Learned deterministic MLP dynamics (reference dynamics_functions/deterministic_mlp.py:5-51).

A weights container: Dense kernels [in,out] + biases, Keras default init
(Glorot-uniform / zeros).  The planning-time forward pass runs on the GPU inside the fused
rollout kernels (MFMA path); training (SystemDynamicsHandler.train) runs through
dynamics_functions/_train_torch.py on the same device."""
import numpy as np

from .. import _lib as L

_ACT = {None: L.ACT_NONE, "linear": L.ACT_NONE, "none": L.ACT_NONE, "tanh": L.ACT_TANH, "relu": L.ACT_RELU,
        "sigmoid": L.ACT_SIGMOID}


def _act_code(a):
    if a is None or isinstance(a, str):
        key = a.lower() if isinstance(a, str) else None
    else:  # a callable such as np.tanh / torch.tanh / tf.math.tanh: resolve by name
        key = getattr(a, "__name__", str(a)).lower()
    if key not in _ACT:
        raise ValueError("unsupported activation %r (supported: tanh, relu, sigmoid, None)" % (a,))
    return _ACT[key]


class DeterministicMLP:
    _bbmpc_dynamics_kind = L.DYN_MLP

    def __init__(self, layers, activation_functions, loss_fn=None, name=None, seed=None):
        if len(activation_functions) != len(layers) - 1:
            raise ValueError("need one activation per Dense layer")
        self.name = name
        self.layer_sizes = [int(v) for v in layers]
        self.activation_codes = [_act_code(a) for a in activation_functions]
        rng = np.random.default_rng(seed)
        self.weights, self.biases = [], []
        for i in range(1, len(layers)):
            lim = np.sqrt(6.0 / (layers[i - 1] + layers[i]))
            self.weights.append(rng.uniform(-lim, lim, size=(layers[i - 1], layers[i])).astype(np.float32))
            self.biases.append(np.zeros((layers[i],), np.float32))
        self.loss_fn = loss_fn
        self._version = 0

    def set_weights(self, weights, biases):
        if len(weights) != len(self.weights):
            raise ValueError("expected %d layers" % len(self.weights))
        for i, (w, b) in enumerate(zip(weights, biases)):
            w = np.asarray(w, np.float32)
            b = np.asarray(b, np.float32)
            if w.shape != self.weights[i].shape or b.shape != self.biases[i].shape:
                raise ValueError("layer %d shape mismatch" % i)
            self.weights[i], self.biases[i] = w.copy(), b.copy()
        self._version += 1

    def save(self, path):
        np.savez(path, n_layers=len(self.weights), layers=np.array(self.layer_sizes),
                 activations=np.array(self.activation_codes),
                 **{"W%d" % i: w for i, w in enumerate(self.weights)},
                 **{"b%d" % i: b for i, b in enumerate(self.biases)})

    @classmethod
    def load(cls, path):
        z = np.load(path)
        inv = {v: k for k, v in _ACT.items() if k in (None, "tanh", "relu", "sigmoid")}
        m = cls(list(z["layers"]), [inv[int(c)] for c in z["activations"]])
        n = int(z["n_layers"])
        m.set_weights([z["W%d" % i] for i in range(n)], [z["b%d" % i] for i in range(n)])
        return m

    def __call__(self, x, train=False):
        """x [B, layers[0]] -> [B, layers[-1]]: the Dense stack (deterministic_mlp.py:27-51; `train` is accepted and
        ignored, as there).  Runs on the GPU through bbmpc_mlp_forward -- the per-row code the control step's tail
        uses; inside rollouts the same weights run fused on the matrix cores."""
        x = np.asarray(x, np.float32)
        eng = self.__dict__.get("_fwd_engine")
        if eng is None or self.__dict__.get("_fwd_version") != self._version:
            from ..engine import Engine
            dim_s = self.layer_sizes[-1]
            dim_u = self.layer_sizes[0] - dim_s
            if dim_u < 1:
                raise ValueError("DeterministicMLP maps [state | action] -> state delta: layers[0] must exceed layers[-1]")
            if eng is None:
                # the reward kind is irrelevant for a forward pass; REW_USER puts no constraint on dim_S
                eng = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_USER, [-1.0] * dim_u, [1.0] * dim_u, dim_s=dim_s, num_agents=1,
                             planning_horizon=1)
                self._fwd_engine = eng
            eng.set_mlp(self.weights, self.biases, self.activation_codes, None)
            self._fwd_version = self._version
        return eng.mlp_forward(x)

    # -- losses (deterministic_mlp.py:53-92: both are loss_fn = Keras MeanSquaredError unless overridden) ----------
    def get_loss(self, expected_output, predictions):
        if self.loss_fn is not None:
            return self.loss_fn(expected_output, predictions)
        d = np.asarray(expected_output, np.float32) - np.asarray(predictions, np.float32)
        return np.float32(np.mean(d * d))

    def get_validation_loss(self, expected_output, predictions):
        return self.get_loss(expected_output, predictions)
