"""torch-CPU restatement of the hot path, one torch op per TensorFlow op -- TEST / BASELINE INFRASTRUCTURE ONLY.

SURVEY.md 8(d)(ii): the reference's TF-2.0 CPU path cannot be timed (TensorFlow is absent here and on the GPU box), so
the closest stand-in for it is the same op graph on another eager CPU tensor framework: fp32, the framework's own
multi-threaded sgemm for the Dense layers (what TF-CPU's Eigen contraction is), one kernel per elementwise op, the
planning loop as a Python loop of H bodies (the reference's `tf.while_loop` body, deterministic.py:62-73, runs its ops
one by one as well).  Random draws are produced INSIDE the timed region by the framework's generators, as the
reference's graph does (cem.py:90, pi2.py:65, random_search.py:40); for parity tests they can be injected instead.

Only `tests/` and `bench.py`'s `cpu_baseline` leg may import this file; the product never does.  PARITY UNPINNED, like
`oracle_np.py`, whose results it is held to in tests/test_oracle_torch.py (same injected draws, fp32 tolerance: torch's
sgemm and libm are not correctly rounded, oracle_np is).

`path:line` citations are relative to /root/reference/blackbox_mpc/.
"""
import math

import torch

F = torch.float32


def _t(x):
    return torch.as_tensor(x, dtype=F)


# ---- leaf math -------------------------------------------------------------------------------------------------------
def pendulum_dynamics(x):
    """PendulumTrueModel.__call__  utils/pendulum.py:58-92 (quirk Q9: unclipped speed integrates the angle)."""
    u, thdot = x[:, 3], x[:, 2]
    theta = torch.atan2(x[:, 1], x[:, 0])                                   # :82
    acc = -15.0 * torch.sin(theta + math.pi) + 3.0 * u                        # :83-84
    newthdot = thdot + acc * 0.05                                             # :85
    newth = theta + newthdot * 0.05                                           # :86
    newthdot = torch.clamp(newthdot, -8.0, 8.0)                               # :87
    new_state = torch.stack([torch.cos(newth), torch.sin(newth), newthdot], dim=1)   # :88-90
    return new_state - x[:, :3]                                               # :91


def pendulum_reward(cur, actions, nxt):
    """pendulum_reward_function  utils/pendulum.py:10-35 as EXECUTED (quirk Q1: called as (cur, actions, next))."""
    th = torch.atan2(cur[:, 1], cur[:, 0])
    ang = torch.remainder(th + math.pi, 2 * math.pi) - math.pi                # :5-7 (floormod)
    first = ang * ang + 0.1 * (cur[:, 2] * cur[:, 2])
    return -first - 0.001 * torch.sum(nxt * nxt, dim=1)


def cheetah_reward(cur, actions, nxt):
    """reward_function  /root/reference/tutorials/mujoco/cost_func.py:5-22."""
    r = torch.zeros((cur.shape[0],), dtype=F)
    r = torch.where(cur[:, 5] >= 0.2, r - 10.0, r)                            # :9-11
    r = torch.where(cur[:, 6] >= 0.0, r - 10.0, r)                            # :13-15
    r = torch.where(cur[:, 7] >= 0.0, r - 10.0, r)                            # :17-19
    r = r + (nxt[:, 17] - cur[:, 17]) / 0.01                                  # :20
    return r - 0.0 * torch.sum(actions * actions, dim=1)                      # :21


REWARDS = {"pendulum": pendulum_reward, "cheetah": cheetah_reward}
ACTS = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid, None: None}


class MLP:
    """DeterministicMLP.__call__  dynamics_functions/deterministic_mlp.py:27-51 (Keras Dense: x @ W + b)."""

    def __init__(self, weights, biases, acts):
        self.w = [_t(w).contiguous() for w in weights]
        self.b = [_t(b) for b in biases]
        self.acts = [ACTS[a] for a in acts]

    def __call__(self, x):
        for w, b, a in zip(self.w, self.b, self.acts):
            x = torch.matmul(x, w) + b                                        # :49-50  MatMul, BiasAdd
            if a is not None:
                x = a(x)
        return x


class Handler:
    """SystemDynamicsHandler.process_input / process_output  dynamics_handlers/system_dynamics_handler.py:97-161."""

    def __init__(self, dynamics, true_model, is_normalized=True, stats=None):
        self.dynamics, self.true_model, self.is_normalized = dynamics, true_model, is_normalized
        if (not true_model) and is_normalized:
            self.mean_s, self.std_s, self.mean_a, self.std_a, self.mean_t, self.std_t = [_t(v) for v in stats]

    def process_input(self, s, a):
        if self.true_model or not self.is_normalized:
            return torch.cat([s, a], dim=-1)
        return torch.cat([(s - self.mean_s) / (self.std_s + 1e-7), (a - self.mean_a) / (self.std_a + 1e-7)], dim=-1)

    def process_output(self, s, raw):
        if self.true_model or not self.is_normalized:
            return raw + s
        return (self.mean_t + raw * (self.std_t + 1e-7)) + s                  # utils/transforms.py:20-34


class Evaluator:
    """DeterministicTrajectoryEvaluator  trajectory_evaluators/deterministic.py:26-127."""

    def __init__(self, reward, handler):
        self.reward, self.handler = REWARDS[reward], handler

    def predict_next_state(self, s, a):                                       # :79-103
        return self.handler.process_output(s, self.handler.dynamics(self.handler.process_input(s, a)))

    def __call__(self, states, seqs):
        n, a, h, u = seqs.shape
        seq = seqs.reshape(n * a, h, u).permute(1, 0, 2)                      # :53-56
        state = states.repeat(n, 1)                                           # :57
        total = torch.zeros((n * a,), dtype=F)
        for t in range(h):                                                    # :62-73
            act = seq[t]
            nxt = self.predict_next_state(state, act)
            total = total + self.reward(state, act, nxt)
            state = nxt
        total = total.reshape(n, a)
        return torch.where(torch.isnan(total), torch.full_like(total, -1e6), total)   # :75-77


# ---- optimizers ------------------------------------------------------------------------------------------------------
def truncated_normal(shape, gen):
    """tf.random.truncated_normal: unit normal re-drawn until |z| < 2."""
    z = torch.randn(shape, dtype=F, generator=gen)
    bad = z.abs() >= 2.0
    while bool(bad.any()):
        z = torch.where(bad, torch.randn(shape, dtype=F, generator=gen), z)
        bad = z.abs() >= 2.0
    return z


class _Base:
    """OptimizerBase  optimizers/optimizer_base.py:6-95 (no exploration noise: the benchmark never asks for it)."""

    def __init__(self, ev, low, high, horizon, num_agents, seed=0):
        self.ev, self.lo, self.hi = ev, _t(low).reshape(-1), _t(high).reshape(-1)
        self.H, self.A, self.U = int(horizon), int(num_agents), self.lo.numel()
        self.lo_h, self.hi_h = self.lo.repeat(self.H, 1), self.hi.repeat(self.H, 1)
        self.gen = torch.Generator().manual_seed(seed)

    def _mean0(self):
        return ((self.lo + self.hi) / 2).repeat(self.A, self.H, 1)

    def _var0(self):
        return (((self.lo - self.hi) ** 2) / 16).repeat(self.A, self.H, 1)

    def call(self, state, noise=None):
        state = _t(state)
        action = self._optimize(state, noise)
        nxt = self.ev.predict_next_state(state, action)                       # :91-94
        return action, nxt, self.ev.reward(state, action, nxt)


class RandomSearch(_Base):
    """optimizers/random_search.py:38-48."""

    def __init__(self, ev, low, high, horizon, population, num_agents, seed=0):
        super().__init__(ev, low, high, horizon, num_agents, seed)
        self.N = population

    def reset(self):
        pass

    def _optimize(self, state, noise):
        shape = (self.N, self.A, self.H, self.U)
        u01 = _t(noise["uniform"]) if noise else torch.rand(shape, dtype=F, generator=self.gen)
        samples = u01 * (self.hi_h - self.lo_h) + self.lo_h                   # :40-41
        rewards = self.ev(state, samples)
        best = torch.argmax(rewards, dim=0)                                   # :43
        return samples[best, torch.arange(self.A), 0, :]                      # :44-47


class CEM(_Base):
    """optimizers/cem.py:46-136 (quirk Q2: no warm start)."""

    def __init__(self, ev, low, high, horizon, max_iterations, population, num_elite, num_agents, alpha=0.25, seed=0):
        super().__init__(ev, low, high, horizon, num_agents, seed)
        self.iters, self.N, self.k, self.alpha = max_iterations, population, num_elite, alpha
        self.prev, self.var0 = self._mean0(), self._var0()

    def reset(self):
        self.prev = self._mean0()

    def _optimize(self, state, noise):
        mean, var = self.prev.clone(), self.var0.clone()
        for it in range(self.iters):
            lb, ub = mean - self.lo_h, self.hi_h - mean                       # :79-80
            cv = torch.minimum(torch.minimum((lb / 2) ** 2, (ub / 2) ** 2), var)   # :81-88
            xi = _t(noise["trunc"][it]) if noise else truncated_normal((self.N, self.A, self.H, self.U), self.gen)
            samples = xi * torch.sqrt(cv) + mean                              # :90-94
            rewards = self.ev(state, samples)                                 # :95-96
            _, idx = torch.topk(rewards.t(), self.k, dim=1, sorted=True)      # :97-99
            st = samples.permute(1, 0, 2, 3)
            elites = torch.stack([st[a][idx[a]] for a in range(self.A)], 0)   # :100-111
            new_mean = elites.mean(dim=1)                                     # :112
            new_var = ((elites - new_mean[:, None]) ** 2).mean(dim=1)         # :113-119
            mean = self.alpha * mean + (1 - self.alpha) * new_mean            # :121-122
            var = self.alpha * var + (1 - self.alpha) * new_var               # :123-125
        return mean[:, 0]


class PI2(_Base):
    """optimizers/pi2.py:41-96 (quirk Q8: constant variance, penalty = norm**2)."""

    def __init__(self, ev, low, high, horizon, max_iterations, population, num_agents, lamda=1.0, seed=0):
        super().__init__(ev, low, high, horizon, num_agents, seed)
        self.iters, self.N, self.lamda = max_iterations, population, lamda
        self.prev, self.var = self._mean0(), self._var0()

    def reset(self):
        self.prev = self._mean0()

    def _optimize(self, state, noise):
        mean = self.prev.clone()
        for it in range(self.iters):
            xi = _t(noise["trunc"][it]) if noise else truncated_normal((self.N, self.A, self.H, self.U), self.gen)
            samples = xi * torch.sqrt(self.var) + mean                        # :65-69
            feas = torch.maximum(torch.minimum(samples, self.hi_h), self.lo_h)    # :70-71
            pen = torch.norm((samples - feas).reshape(self.N, self.A, -1), dim=2) ** 2   # :72-75
            rewards = self.ev(state, feas) - pen                              # :77
            costs = (-rewards).t()                                            # :78-79
            beta = costs.min(dim=1).values                                    # :81
            prob = torch.exp(-(1.0 / self.lamda) * (costs - beta[:, None]))   # :82
            omega = (1.0 / prob.sum(dim=1))[:, None] * prob                   # :83-85
            mean = (feas.permute(1, 0, 2, 3) * omega[:, :, None, None]).sum(dim=1)   # :86-87
        self.prev = torch.cat([mean[:, 1:], mean[:, -1:]], dim=1)             # :92-93
        return mean[:, 0]


def make(opt, env, lo, hi, N, A, H, iters, k, mlp=None, stats=None, seed=0):
    """The optimizer of a bench configuration: env 'pendulum' (true model) or 'cheetah' (mlp = (weights, biases, acts))."""
    if env == "pendulum":
        ev = Evaluator("pendulum", Handler(pendulum_dynamics, True))
    else:
        ev = Evaluator("cheetah", Handler(MLP(*mlp), False, True, stats))
    if opt == "RandomSearch":
        return RandomSearch(ev, lo, hi, H, N, A, seed)
    if opt == "CEM":
        return CEM(ev, lo, hi, H, iters, N, k, A, seed=seed)
    if opt == "PI2":
        return PI2(ev, lo, hi, H, iters, N, A, seed=seed)
    raise ValueError(opt)
