"""CPU oracle for the dynamics-model training row -- TEST INFRASTRUCTURE ONLY.

NumPy restatement of what the reference does when `SystemDynamicsHandler.train` is called
(dynamics_handlers/system_dynamics_handler.py:163-349, paths relative to /root/reference/blackbox_mpc/):
dataset assembly from episode lists, train/validation split, freeze-after-first normalisation statistics,
normalisation, shuffled drop-remainder mini-batches, MSE loss (deterministic_mlp.py:53-92, Keras
MeanSquaredError), one Keras-Adam step per batch, per-epoch mean losses.

PARITY UNPINNED (see oracle_np.py): the reference ships no tests and TensorFlow is not installable here.  The third
party arithmetic on this row is `tf.keras.optimizers.Adam` (tensorflow==2.0.0, setup.py:12) -- restated below from
its published update rule (Kingma & Ba Alg. 1 in the "epsilon outside" form Keras documents):
    lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
    w   -= lr_t * m / (sqrt(v) + eps),        b1 = 0.9, b2 = 0.999, eps = 1e-7
and `np.random.choice` / `tf.data.Dataset.shuffle`, whose draws are *injected* here (split mask, one permutation
per epoch) exactly as the hot-path oracle injects its noise.

Arithmetic is float64 throughout: a neutral reference for any fp32 summation order; the product (PyTorch on the
GPU, fp32) is compared within a stated tolerance.
"""
import numpy as np

ACTS = {"tanh": (np.tanh, lambda y: 1.0 - y * y),
        "relu": (lambda x: np.maximum(x, 0.0), lambda y: (y > 0).astype(np.float64)),
        "sigmoid": (lambda x: 1.0 / (1.0 + np.exp(-x)), lambda y: y * (1.0 - y)),
        None: (lambda x: x, lambda y: np.ones_like(y))}


def assemble_dataset(observations_trajectories, actions_trajectories):
    """_append_to_training_dataset :292-310: episodes obs [T+1,A,S], acs [T,A,U] -> rows (s_t, a_t) and
    targets s_{t+1} - s_t (utils/transforms.py default_transform_targets), episode-major, then agent, then t."""
    acs_all = np.array(actions_trajectories)
    num_agents = acs_all.shape[2]
    rows_in, rows_out = [], []
    for obs, acs in zip(observations_trajectories, acs_all):
        obs = np.asarray(obs)
        for agent in range(num_agents):
            states = obs[:-1, agent]
            rows_in.append(np.concatenate([states, acs[:, agent]], axis=-1))
            rows_out.append(obs[1:, agent] - states)
    d_in = np.array(rows_in, dtype=np.float32)
    d_in = d_in.reshape(-1, d_in.shape[-1])
    d_out = np.array(rows_out, dtype=np.float32)
    d_out = d_out.reshape(-1, d_out.shape[-1])
    return d_in, d_out


def normalization_stats(train_in, train_out, dim_s):
    """_recompute_normalization :340-349 (np.mean / np.std, population std, over the TRAINING rows only)."""
    return [np.mean(train_in[:, :dim_s], axis=0), np.std(train_in[:, :dim_s], axis=0),
            np.mean(train_in[:, dim_s:], axis=0), np.std(train_in[:, dim_s:], axis=0),
            np.mean(train_out, axis=0), np.std(train_out, axis=0)]


def normalize(d_in, d_out, stats, dim_s):
    """_normalize_data :333-338"""
    ms, ss, ma, sa, mt, st = stats
    s = (d_in[:, :dim_s] - ms) / (ss + 1e-7)
    a = (d_in[:, dim_s:] - ma) / (sa + 1e-7)
    t = (d_out - mt) / (st + 1e-7)
    return np.concatenate([s, a], axis=1), t


def forward(weights, biases, acts, x):
    ys = [np.asarray(x, np.float64)]
    for w, b, a in zip(weights, biases, acts):
        ys.append(ACTS[a][0](ys[-1] @ w + b))
    return ys


def mse_and_grads(weights, biases, acts, x, y):
    """MeanSquaredError (mean over every element) + backprop through the Dense stack."""
    ys = forward(weights, biases, acts, x)
    diff = ys[-1] - y
    loss = float(np.mean(diff * diff))
    g = 2.0 * diff / diff.size
    gw, gb = [None] * len(weights), [None] * len(weights)
    for l in reversed(range(len(weights))):
        g = g * ACTS[acts[l]][1](ys[l + 1])
        gw[l] = ys[l].T @ g
        gb[l] = g.sum(axis=0)
        g = g @ weights[l].T
    return loss, gw, gb


class KerasAdam:
    def __init__(self, params, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t = learning_rate, beta_1, beta_2, epsilon, 0
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = self.b1 * self.m[i] + (1.0 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1.0 - self.b2) * g * g
            p -= lr_t * self.m[i] / (np.sqrt(self.v[i]) + self.eps)


class KerasSGD:
    """tf.keras.optimizers.SGD(learning_rate) with TF-2.0 defaults: momentum 0."""

    def __init__(self, params, learning_rate=1e-2):
        self.lr = learning_rate

    def step(self, params, grads):
        for p, g in zip(params, grads):
            p -= self.lr * g


class KerasRMSprop:
    """tf.keras.optimizers.RMSprop(learning_rate) with TF-2.0 defaults: rho 0.9, momentum 0, epsilon 1e-7, not centered."""

    def __init__(self, params, learning_rate=1e-3, rho=0.9, epsilon=1e-7):
        self.lr, self.rho, self.eps = learning_rate, rho, epsilon
        self.v = [np.zeros_like(p) for p in params]

    def step(self, params, grads):
        for i, (p, g) in enumerate(zip(params, grads)):
            self.v[i] = self.rho * self.v[i] + (1.0 - self.rho) * g * g
            p -= self.lr * g / (np.sqrt(self.v[i]) + self.eps)


OPTIMIZERS = {"adam": KerasAdam, "sgd": KerasSGD, "rmsprop": KerasRMSprop}


def train(weights, biases, acts, train_in, train_out, val_in, val_out, permutations, batch_size=128,
          learning_rate=1e-3, rule="adam"):
    """_training_algorithm :243-290.  Inputs are already normalised.  permutations: one index permutation of the
    training rows per epoch (tf.data shuffle(buffer = all rows) reshuffles every epoch); batches drop the remainder.
    A fresh Adam is created per call, as the reference does (:258).  Returns (weights, biases, train_loss[epochs],
    val_loss[epochs])."""
    w = [np.array(x, np.float64) for x in weights]
    b = [np.array(x, np.float64) for x in biases]
    opt = OPTIMIZERS[rule](w + b, learning_rate)
    tl, vl = [], []
    for perm in permutations:
        nb, acc = 0, 0.0
        for s in range(0, len(perm) - batch_size + 1, batch_size):
            idx = perm[s:s + batch_size]
            loss, gw, gb = mse_and_grads(w, b, acts, train_in[idx], train_out[idx])
            opt.step(w + b, gw + gb)
            acc += loss
            nb += 1
        tl.append(acc / nb if nb else np.nan)          # the reference divides by zero batches the same way
        nb, acc = 0, 0.0
        for s in range(0, val_in.shape[0] - batch_size + 1, batch_size):
            pred = forward(w, b, acts, val_in[s:s + batch_size])[-1]
            d = pred - val_out[s:s + batch_size]
            acc += float(np.mean(d * d))
            nb += 1
        vl.append(acc / nb if nb else np.nan)
    return w, b, np.array(tl), np.array(vl)
