"""On-device training of the Dense stack: the counterpart of the reference's
`SystemDynamicsHandler._training_algorithm` (dynamics_handlers/system_dynamics_handler.py:243-290) with
`DeterministicMLP.get_loss` (dynamics_functions/deterministic_mlp.py:53-92, Keras MeanSquaredError) and
`tf.keras.optimizers.Adam` (TF 2.0: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), w -= lr_t*m/(sqrt(v)+1e-7)).

MI355X shape of it: the whole (normalised) dataset lives in HBM, the epoch permutation is drawn on the device, and
one training step -- batch gather, forward, hand-written backward, Adam -- is ~40 tiny kernels on a 26-200-200-20
network, i.e. launch bound: the step is captured once in a HIP graph and replayed per batch (no autograd tape, no
host synchronisation inside an epoch; the loss is accumulated on the device and read once per epoch).
PyTorch-ROCm is plumbing here (hipBLASLt GEMMs, streams, graphs)."""
import os

import numpy as np

from .. import _lib as L


def _act(code, x):
    import torch
    if code == L.ACT_TANH:
        return torch.tanh(x)
    if code == L.ACT_RELU:
        return torch.relu(x)
    if code == L.ACT_SIGMOID:
        return torch.sigmoid(x)
    return x


def _act_grad(code, y, g):
    if code == L.ACT_TANH:
        return g * (1.0 - y * y)
    if code == L.ACT_RELU:
        return g * (y > 0).to(g.dtype)
    if code == L.ACT_SIGMOID:
        return g * (y * (1.0 - y))
    return g


class DenseTrainer:
    def __init__(self, weights, biases, act_codes, device, learning_rate=1e-3, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7, rule="adam", rho=0.9):
        """rule: "adam" | "sgd" | "rmsprop" -- tf.keras.optimizers.{Adam, SGD, RMSprop}(learning_rate=lr) with their
        TF-2.0 defaults, which is how the reference instantiates `nn_optimizer` (system_dynamics_handler.py:261):
          sgd      w -= lr * g                                              (momentum 0)
          rmsprop  v = rho*v + (1-rho)*g^2 ;  w -= lr * g / (sqrt(v) + eps)   (rho 0.9, momentum 0, eps 1e-7, not centered)"""
        if rule not in ("adam", "sgd", "rmsprop"):
            raise ValueError("unknown optimizer rule %r" % (rule,))
        self.rule, self.rho = rule, float(rho)
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        self.acts = list(act_codes)
        self.w = [torch.tensor(np.asarray(w, np.float32), device=self.dev) for w in weights]
        self.b = [torch.tensor(np.asarray(b, np.float32), device=self.dev) for b in biases]
        self.params = self.w + self.b
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        # running powers b1^t, b2^t as device scalars so that the step is graph-capturable
        self.b1t = torch.ones((), device=self.dev, dtype=torch.float64)
        self.b2t = torch.ones((), device=self.dev, dtype=torch.float64)
        self.loss_acc = torch.zeros((), device=self.dev, dtype=torch.float32)
        self._graph = None

    # -- one step on (x, y): forward, MSE, backward, Keras-Adam ---------------------------------------------------
    def forward(self, x):
        ys = [x]
        for w, b, a in zip(self.w, self.b, self.acts):
            ys.append(_act(a, self.torch.addmm(b, ys[-1], w)))
        return ys

    def _step(self, x, y):
        torch = self.torch
        ys = self.forward(x)
        diff = ys[-1] - y
        self.loss_acc += (diff * diff).mean()
        g = diff * (2.0 / diff.numel())
        n = len(self.w)
        grads = [None] * (2 * n)
        for l in reversed(range(n)):
            g = _act_grad(self.acts[l], ys[l + 1], g)
            grads[l] = ys[l].t() @ g
            grads[n + l] = g.sum(dim=0)
            if l:
                g = g @ self.w[l].t()
        if self.rule == "sgd":
            torch._foreach_add_(self.params, grads, alpha=-self.lr)
            return
        if self.rule == "rmsprop":
            torch._foreach_mul_(self.v, self.rho)
            torch._foreach_addcmul_(self.v, grads, grads, value=1.0 - self.rho)
            den = torch._foreach_sqrt(self.v)
            torch._foreach_add_(den, self.eps)
            upd = torch._foreach_div(grads, den)
            torch._foreach_add_(self.params, upd, alpha=-self.lr)
            return
        self.b1t *= self.b1
        self.b2t *= self.b2
        lr_t = (self.lr * torch.sqrt(1.0 - self.b2t) / (1.0 - self.b1t)).to(torch.float32)
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, grads, alpha=1.0 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, grads, grads, value=1.0 - self.b2)
        den = torch._foreach_sqrt(self.v)
        torch._foreach_add_(den, self.eps)
        upd = torch._foreach_div(self.m, den)
        torch._foreach_mul_(upd, -lr_t)
        torch._foreach_add_(self.params, upd)

    def _gather_step(self):
        # batch = rows perm[pos : pos + B]; `pos` is a device scalar advanced here so that an epoch is nothing but
        # graph replays
        idx = self._perm.index_select(0, self._ar + self._pos)
        self._pos += self._ar.shape[0]
        self._step(self._din.index_select(0, idx), self._dout.index_select(0, idx))

    def fit(self, train_in, train_out, val_in, val_out, epochs, batch_size, permutations=None, generator_seed=None):
        """Returns (train_loss[epochs], val_loss[epochs]); weights are updated in place (fetch with `numpy_params`)."""
        torch = self.torch
        self._din = torch.as_tensor(np.ascontiguousarray(train_in, np.float32)).to(self.dev)
        self._dout = torch.as_tensor(np.ascontiguousarray(train_out, np.float32)).to(self.dev)
        vin = torch.as_tensor(np.ascontiguousarray(val_in, np.float32)).to(self.dev)
        vout = torch.as_tensor(np.ascontiguousarray(val_out, np.float32)).to(self.dev)
        n = self._din.shape[0]
        nb = n // batch_size                                    # drop_remainder=True (:201-202)
        self._perm = torch.zeros((max(n, batch_size),), dtype=torch.int64, device=self.dev)
        self._ar = torch.arange(batch_size, dtype=torch.int64, device=self.dev)
        self._pos = torch.zeros((), dtype=torch.int64, device=self.dev)
        gen = None
        if permutations is None:
            gen = torch.Generator(device=self.dev)
            gen.manual_seed(int(generator_seed) if generator_seed is not None else int.from_bytes(os.urandom(4), "little"))
        use_graph = self.dev.type == "cuda" and nb > 0 and os.environ.get("BBMPC_TRAIN_GRAPH", "1") != "0"
        self._graph = None                      # the graph bakes in this call's dataset / index buffers: capture per fit
        if use_graph:
            # warm-up on a side stream (allocator + hipBLASLt workspaces), state restored afterwards, then capture
            state = self.params + self.m + self.v + [self.b1t, self.b2t, self.loss_acc, self._pos]
            snap = [t.clone() for t in state]
            s = torch.cuda.Stream(device=self.dev)
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                for _ in range(3):
                    self._pos.zero_()
                    self._gather_step()
            torch.cuda.current_stream(self.dev).wait_stream(s)
            self._graph = torch.cuda.CUDAGraph()
            self._pos.zero_()
            with torch.cuda.graph(self._graph):
                self._gather_step()
            for t, c in zip(state, snap):
                t.copy_(c)
        tl, vl = np.full((epochs,), np.nan), np.full((epochs,), np.nan)
        for e in range(epochs):
            if permutations is not None:
                perm = torch.as_tensor(np.asarray(permutations[e], np.int64)).to(self.dev)
            else:
                perm = torch.randperm(n, generator=gen, device=self.dev)
            self.loss_acc.zero_()
            self._perm[:n].copy_(perm)
            self._pos.zero_()
            for bi in range(nb):
                if use_graph:
                    self._graph.replay()
                else:
                    self._gather_step()
            if nb:
                tl[e] = float(self.loss_acc.item()) / nb
            nvb = vin.shape[0] // batch_size
            if nvb:
                pv = self.forward(vin[:nvb * batch_size])[-1]
                d = (pv - vout[:nvb * batch_size]).reshape(nvb, -1)
                vl[e] = float((d * d).mean(dim=1).mean().item())    # mean of per-batch MSEs (:276-284)
        return tl, vl

    def numpy_params(self):
        return [w.detach().cpu().numpy() for w in self.w], [b.detach().cpu().numpy() for b in self.b]
