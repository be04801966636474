"""CPU oracle for the sampling-MPC hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy fp32 restatement, op for op, of the arithmetic the
reference (ossamaAhmed/blackbox_mpc v0.3, TensorFlow 2.0 Python) executes on
its hot path  MPCPolicy.act -> Optimizer -> TrajectoryEvaluator -> dynamics /
reward.  It exists so that the HIP engine in ``blackbox_mpc_amd`` can be
checked for results parity.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product never does.

PARITY UNPINNED.  The reference ships no tests / golden vectors, never seeds
its RNG, and cannot be imported here (TensorFlow 2.0 is absent and cannot be
installed), so this oracle cannot be pinned against outputs of the reference
itself.  What pins it instead (see tests/test_oracle_kat.py):
  * hand-derived known-answer values for the pendulum model / reward,
  * closed-form refit checks (CEM elites / PI2 softmin / argmax tie rules),
  * CMA-ES constructor constants,
each derived independently from the reference source lines cited below.

Conventions
-----------
* Every tensor is float32 and every TF op is one NumPy op with fp32 rounding
  (no fused multiply-add).  Transcendentals (sin, cos, atan2, exp, tanh, pow,
  log) are evaluated in float64 and rounded once to fp32, i.e. the oracle is
  the correctly-rounded fp32 function; TF-CPU (Eigen) and the HIP engine are
  each within ~1-2 ulp of it.  Dense layers accumulate in float64 and round
  once (a neutral reference for any fp32 accumulation order).
* RNG: TF's Philox streams are third-party and unseeded in the reference, so
  every random draw is *injected* as a standard-noise tensor:
  unit normal truncated to |z|<2 (``tf.random.truncated_normal``), U[0,1)
  (``tf.random.uniform``), N(0,1) (``tf.random.normal``), Rademacher +-1.
  All implementations (oracle, HIP) consume identical draws.
* Layouts are the reference's: samples [N, A, H, U], rewards [N, A],
  states [A, S].

All ``path:line`` citations are relative to /root/reference/blackbox_mpc/.
"""
import numpy as np

F = np.float32
PI32 = F(np.pi)
TWO_PI32 = F(2 * np.pi)


def f32(x):
    return np.asarray(x, dtype=np.float32)


def _via64(fn, *xs):
    """Correctly-rounded fp32 transcendental: evaluate in fp64, round once."""
    with np.errstate(all="ignore"):
        return fn(*[np.asarray(x, dtype=np.float64) for x in xs]).astype(np.float32)


def sin32(x):
    return _via64(np.sin, x)


def cos32(x):
    return _via64(np.cos, x)


def atan2_32(y, x):
    return _via64(np.arctan2, y, x)


def exp32(x):
    return _via64(np.exp, x)


def tanh32(x):
    return _via64(np.tanh, x)


def log32(x):
    return _via64(np.log, x)


def pow32(x, y):
    return _via64(np.power, x, y)


def sqrt32(x):
    with np.errstate(all="ignore"):
        return np.sqrt(f32(x))  # IEEE correctly rounded in fp32


def floormod32(x, y):
    """TF FloorMod for floats: fmod, then shift into the divisor's sign.
    Used by utils/pendulum.py:7 (`%` on a tf.float32 tensor)."""
    x = f32(x)
    y = f32(y)
    with np.errstate(all="ignore"):
        r = np.fmod(x, y)
        fix = (r != 0) & ((y < 0) != (r < 0))
        return np.where(fix, (r + y).astype(np.float32), r).astype(np.float32)


def seq_sum(x, axis):
    """Strictly sequential fp32 sum along `axis` (index 0,1,2,... order)."""
    x = f32(x)
    x = np.moveaxis(x, axis, 0)
    acc = np.zeros(x.shape[1:], dtype=np.float32)
    for i in range(x.shape[0]):
        acc = (acc + x[i]).astype(np.float32)
    return acc


# ---------------------------------------------------------------------------
# Leaf math: analytic pendulum model + rewards
# ---------------------------------------------------------------------------
def pendulum_dynamics(x):
    """PendulumTrueModel.__call__  utils/pendulum.py:58-92.
    x [B,4] = (cos th, sin th, thdot, u)  ->  delta [B,3] (NOT the next state).
    Quirk Q9: th is integrated with the UNCLIPPED new speed (:86) and the
    torque is never clipped (max_torque unused)."""
    x = f32(x)
    u = x[:, 3]
    thdot = x[:, 2]
    theta = atan2_32(x[:, 1], x[:, 0])                              # :82
    # -3*g/(2*l) = (-3*10)/(2*1) = -15 ;  3/(m*l**2) = 3  (all exact in fp32)
    acc = (F(-15.0) * sin32((theta + PI32).astype(F))).astype(F)    # :83
    acc = (acc + (F(3.0) * u).astype(F)).astype(F)                  # :83-84
    newthdot = (thdot + (acc * F(0.05)).astype(F)).astype(F)        # :85
    newth = (theta + (newthdot * F(0.05)).astype(F)).astype(F)      # :86
    newthdot = np.clip(newthdot, F(-8.0), F(8.0))                   # :87
    new_state = np.stack([cos32(newth), sin32(newth), newthdot], axis=1)  # :88-90
    return (new_state - x[:, :3]).astype(F)                         # :91


def pendulum_reward(current_state, actions, next_state, as_executed=True):
    """pendulum_reward_function  utils/pendulum.py:10-35.

    Declared (current_state, next_state, actions) but the evaluator calls it
    positionally as (cur, actions, next)  trajectory_evaluators/deterministic.py:65-66,
    :126-127  => quirk Q1: the 'action cost' term is 0.001*sum(next_state**2).
    `as_executed=False` gives the intended (declared) order instead."""
    cur = f32(current_state)
    act_term_src = f32(next_state) if as_executed else f32(actions)
    th = atan2_32(cur[:, 1], cur[:, 0])
    ang = (floormod32((th + PI32).astype(F), TWO_PI32) - PI32).astype(F)   # :5-7
    a2 = (ang * ang).astype(F)                                              # **2
    v2 = (cur[:, 2] * cur[:, 2]).astype(F)
    first = (a2 + (F(0.1) * v2).astype(F)).astype(F)
    sq = (act_term_src * act_term_src).astype(F)
    ssum = seq_sum(sq, axis=1)
    return ((-first).astype(F) - (F(0.001) * ssum).astype(F)).astype(F)


def cheetah_reward(current_state, actions, next_state):
    """reward_function  /root/reference/tutorials/mujoco/cost_func.py:5-22
    (HalfCheetahEnvModified obs, S=20: index 17 = torso COM x)."""
    cur = f32(current_state)
    nxt = f32(next_state)
    act = f32(actions)
    r = np.zeros((cur.shape[0],), dtype=F)
    r = np.where(cur[:, 5] >= F(0.2), (r + F(-10.0)).astype(F), r)          # :9-11
    r = np.where(cur[:, 6] >= F(0.0), (r + F(-10.0)).astype(F), r)          # :13-15
    r = np.where(cur[:, 7] >= F(0.0), (r + F(-10.0)).astype(F), r)          # :17-19
    r = (r + ((nxt[:, 17] - cur[:, 17]).astype(F) / F(0.01)).astype(F)).astype(F)   # :20
    r = (r - (F(0.0) * seq_sum((act * act).astype(F), axis=1)).astype(F)).astype(F)  # :21
    return r.astype(F)


REWARDS = {"pendulum": pendulum_reward, "cheetah": cheetah_reward}


# ---------------------------------------------------------------------------
# Learned dynamics: Dense stack + normalising handler
# ---------------------------------------------------------------------------
class MLP:
    """DeterministicMLP.__call__  dynamics_functions/deterministic_mlp.py:27-51:
    x = act_i(x @ W_i + b_i), W_i is [in, out] (Keras Dense kernel layout).
    acts: list of 'tanh' | 'relu' | 'sigmoid' | None."""

    def __init__(self, weights, biases, acts):
        self.weights = [f32(w) for w in weights]
        self.biases = [f32(b) for b in biases]
        self.acts = list(acts)

    def __call__(self, x):
        x = f32(x)
        for w, b, a in zip(self.weights, self.biases, self.acts):
            y = (x.astype(np.float64) @ w.astype(np.float64)).astype(F)
            y = (y + b).astype(F)
            if a == "tanh":
                y = tanh32(y)
            elif a == "relu":
                y = np.maximum(y, F(0))
            elif a == "sigmoid":
                y = (F(1) / (F(1) + exp32(-y)).astype(F)).astype(F)
            elif a is not None:
                raise ValueError(a)
            x = y
        return x


class Handler:
    """Inference half of SystemDynamicsHandler:
    process_input  dynamics_handlers/system_dynamics_handler.py:97-126,
    process_output :128-161, default_inverse_transform_targets utils/transforms.py:20-34."""

    def __init__(self, dynamics, true_model, is_normalized=True, stats=None):
        self.dynamics = dynamics
        self.true_model = true_model
        self.is_normalized = is_normalized
        if (not true_model) and is_normalized:
            self.mean_s, self.std_s, self.mean_a, self.std_a, self.mean_t, self.std_t = \
                [f32(v) for v in stats]

    def process_input(self, s, a):
        s, a = f32(s), f32(a)
        if self.true_model or not self.is_normalized:
            return np.concatenate([s, a], axis=-1)
        ns = ((s - self.mean_s).astype(F) / (self.std_s + F(1e-7)).astype(F)).astype(F)
        na = ((a - self.mean_a).astype(F) / (self.std_a + F(1e-7)).astype(F)).astype(F)
        return np.concatenate([ns, na], axis=-1)

    def process_output(self, s, raw):
        s, raw = f32(s), f32(raw)
        if self.true_model or not self.is_normalized:
            dev = raw
        else:
            dev = (self.mean_t + (raw * (self.std_t + F(1e-7)).astype(F)).astype(F)).astype(F)
        return (dev + s).astype(F)          # transforms.py:34  delta + current_state


class Evaluator:
    """DeterministicTrajectoryEvaluator  trajectory_evaluators/deterministic.py:26-127."""

    def __init__(self, reward, handler):
        self.reward = REWARDS[reward] if isinstance(reward, str) else reward
        self.handler = handler

    def predict_next_state(self, s, a):                           # :79-103
        x = self.handler.process_input(s, a)
        raw = self.handler.dynamics(x)
        return self.handler.process_output(s, raw)

    def evaluate_next_reward(self, cur, nxt, act):                # :105-127
        return self.reward(cur, act, nxt)

    def __call__(self, current_states, action_sequences, return_final_state=False):
        cs = f32(current_states)
        seq = f32(action_sequences)
        n, a, h, u = seq.shape
        seq = seq.reshape(n * a, h, u).transpose(1, 0, 2)          # :53-56 row b = n*A + a
        state = np.tile(cs, (n, 1))                                # :57
        total = np.zeros((n * a,), dtype=F)
        for t in range(h):                                         # :62-73
            act = seq[t]
            nxt = self.predict_next_state(state, act)
            total = (total + self.reward(state, act, nxt)).astype(F)
            state = nxt
        total = total.reshape(n, a)
        total = np.where(np.isnan(total), F(-1e6), total).astype(F)    # :75-77
        if return_final_state:
            return total, state.reshape(n, a, -1)
        return total


# ---------------------------------------------------------------------------
# Selection helpers with TF tie semantics
# ---------------------------------------------------------------------------
def topk_desc(values, k):
    """tf.nn.top_k(sorted=True) / tf.argsort(DESCENDING, stable): larger first,
    ties -> lower index first.  values [..., N] -> indices [..., k]."""
    v = f32(values)
    n = v.shape[-1]
    idx = np.broadcast_to(np.arange(n), v.shape)
    # lexsort: last key primary.  primary = -value (desc), secondary = index asc.
    order = np.lexsort((idx, -v.astype(np.float64)), axis=-1)
    return order[..., :k]


def argmax_first(values, axis=0):
    """tf.math.argmax: first maximum wins."""
    return np.argmax(f32(values), axis=axis)


# ---------------------------------------------------------------------------
# Optimizers (every random draw injected)
# ---------------------------------------------------------------------------
class OptimizerBase:
    """OptimizerBase  optimizers/optimizer_base.py:6-95."""

    def __init__(self, evaluator, low, high, horizon, num_agents, max_iterations):
        self.ev = evaluator
        self.lo = f32(low).reshape(-1)
        self.hi = f32(high).reshape(-1)
        self.U = self.lo.shape[0]
        self.H = int(horizon)
        self.A = int(num_agents)
        self.iters = max_iterations
        self.lo_h = np.tile(self.lo[None], (self.H, 1))            # :37-42
        self.hi_h = np.tile(self.hi[None], (self.H, 1))
        self.expl_var = (((self.lo - self.hi) ** 2).astype(F) / F(16) * F(0.05)).astype(F)  # :46-48
        self.expl_mean = ((self.hi + self.lo).astype(F) / F(2)).astype(F)                    # :49-50
        self.trace = []

    def _init_mean(self):
        m = ((self.lo + self.hi).astype(F) / F(2)).astype(F)
        return np.tile(m, (self.A, self.H, 1)).astype(F)

    def _init_var(self):
        v = (((self.lo - self.hi) ** 2).astype(F) / F(16)).astype(F)
        return np.tile(v, (self.A, self.H, 1)).astype(F)

    def call(self, state, noise, exploration_noise=None):
        """__call__ :55-95.  exploration_noise: None or unit-truncated-normal [A,U]
        (quirk Q7: its mean is the bounds midpoint, not 0)."""
        state = f32(state)
        action = self._optimize(state, noise)
        if exploration_noise is not None:
            nz = (f32(exploration_noise) * sqrt32(self.expl_var)).astype(F)
            nz = (nz + self.expl_mean).astype(F)
            action = np.clip((action + nz).astype(F), self.lo, self.hi)
        nxt = self.ev.predict_next_state(state, action)
        rew = self.ev.evaluate_next_reward(state, nxt, action)
        return action.astype(F), nxt, rew

    def _clip_h(self, x):
        return np.clip(x, self.lo_h, self.hi_h).astype(F)

    def _penalty(self, x, xf):
        """tf.norm(reshape(x - xf, [N, A, -1]), axis=2) ** 2  (pi2.py:72-75 etc.)."""
        d = (x - xf).astype(F).reshape(x.shape[0], x.shape[1], -1)
        s = seq_sum((d * d).astype(F), axis=2)
        nrm = sqrt32(s)
        return (nrm * nrm).astype(F)


class RandomSearch(OptimizerBase):
    """RandomSearchOptimizer._optimize  optimizers/random_search.py:38-48.
    noise: {'uniform': U[0,1) [N,A,H,U]}."""

    def __init__(self, evaluator, low, high, horizon=50, population=1024, num_agents=5):
        super().__init__(evaluator, low, high, horizon, num_agents, None)
        self.N = population

    def reset(self):
        return

    def _optimize(self, state, noise):
        u01 = f32(noise["uniform"])
        samples = ((u01 * (self.hi_h - self.lo_h).astype(F)).astype(F) + self.lo_h).astype(F)  # :40-41
        rewards = self.ev(state, samples)
        best = argmax_first(rewards, axis=0)                        # :43
        action = samples[best, np.arange(self.A), 0, :]             # :44-47
        self.trace = [dict(samples=samples, rewards=rewards, best=best)]
        return action


class CEM(OptimizerBase):
    """CEMOptimizer  optimizers/cem.py:46-136.
    noise: {'trunc': [iters][N,A,H,U]}.  Quirk Q2: no warm start (assign is
    commented out :133-134) and epsilon is unused.
    `forced_elites`: optional [iters][A,k] index override, or a callable
    (iteration, rewards[N,A], own_topk[A,k]) -> [A,k], used by parity tests to
    keep lock-step when two near-tied rewards swap at the elite boundary."""

    def __init__(self, evaluator, low, high, horizon=50, max_iterations=5, population=500,
                 num_elite=50, num_agents=5, alpha=0.25):
        super().__init__(evaluator, low, high, horizon, num_agents, max_iterations)
        self.N, self.k, self.alpha = population, num_elite, F(alpha)
        self.prev = self._init_mean()
        self.var0 = self._init_var()

    def reset(self):                                               # :138-149
        self.prev = self._init_mean()

    def _optimize(self, state, noise, forced_elites=None):
        mean, var = self.prev.copy(), self.var0.copy()
        self.trace = []
        for it in range(self.iters):
            lb = (mean - self.lo_h).astype(F)                                  # :79
            ub = (self.hi_h - mean).astype(F)                                  # :80
            cv = np.minimum(np.minimum(((lb / F(2)) ** 2).astype(F),
                                       ((ub / F(2)) ** 2).astype(F)), var)     # :81-88
            xi = f32(noise["trunc"][it])
            samples = ((xi * sqrt32(cv)).astype(F) + mean).astype(F)           # :90-94
            rewards = self.ev(state, samples)                                  # :95-96
            idx = topk_desc(rewards.T, self.k)                                 # :97-99  [A,k]
            if callable(forced_elites):
                idx = np.asarray(forced_elites(it, rewards, idx))
            elif forced_elites is not None:
                idx = np.asarray(forced_elites[it])
            st = samples.transpose(1, 0, 2, 3)                                 # [A,N,H,U]
            elites = np.stack([st[a][idx[a]] for a in range(self.A)], 0)       # :100-111 [A,k,H,U]
            new_mean = (seq_sum(elites, axis=1) / F(self.k)).astype(F)         # :112
            dev = (elites - new_mean[:, None]).astype(F)
            new_var = (seq_sum((dev * dev).astype(F), axis=1) / F(self.k)).astype(F)  # :113-119
            one_m = (F(1) - self.alpha).astype(F)
            mean = ((self.alpha * mean).astype(F) + (one_m * new_mean).astype(F)).astype(F)  # :121-122
            var = ((self.alpha * var).astype(F) + (one_m * new_var).astype(F)).astype(F)     # :123-125
            self.trace.append(dict(samples=samples, rewards=rewards, elites=idx,
                                   mean=mean.copy(), var=var.copy(), cvar=cv))
        return mean[:, 0]                                                      # :135


class PI2(OptimizerBase):
    """PI2Optimizer  optimizers/pi2.py:41-96.  noise: {'trunc': [iters][N,A,H,U]}.
    Quirk Q8: variance is constant; penalty is (tf.norm)**2."""

    def __init__(self, evaluator, low, high, horizon=50, max_iterations=5, population=500,
                 num_agents=5, lamda=1.0):
        super().__init__(evaluator, low, high, horizon, num_agents, max_iterations)
        self.N, self.lamda = population, F(lamda)
        self.prev = self._init_mean()
        self.var = self._init_var()

    def reset(self):                                               # :98-105
        self.prev = self._init_mean()

    def _optimize(self, state, noise, rewards_override=None):
        """`rewards_override`: optional callable (iteration, rewards[N,A]) -> rewards[N,A] for lock-step parity tests
        (same contract as PSO._optimize): it checks this oracle's rewards against the device's within the stated
        tolerance and returns the device's, so the exp-weighted mean that follows is compared on identical inputs."""
        mean = self.prev.copy()
        self.trace = []
        for it in range(self.iters):
            xi = f32(noise["trunc"][it])
            samples = ((xi * sqrt32(self.var)).astype(F) + mean).astype(F)     # :65-69
            feas = self._clip_h(samples)                                       # :70-71
            pen = self._penalty(samples, feas)                                 # :72-75
            rewards = (self.ev(state, feas) - pen).astype(F)                   # :77
            if rewards_override is not None:
                rewards = f32(rewards_override(it, rewards))
            costs = (-rewards).T                                               # :78-79 [A,N]
            beta = costs.min(axis=1)                                           # :81
            inv = (F(1) / self.lamda).astype(F)
            prob = exp32(((-inv).astype(F) * (costs - beta[:, None]).astype(F)).astype(F))  # :82
            eta = prob.sum(axis=1, dtype=F)                                    # :83
            omega = ((F(1) / eta).astype(F)[:, None] * prob).astype(F)         # :85
            st = feas.transpose(1, 0, 2, 3)                                    # :86 [A,N,H,U]
            mean = (st * omega[:, :, None, None]).astype(F).sum(axis=1, dtype=F)   # :87
            self.trace.append(dict(samples=feas, rewards=rewards, penalty=pen,
                                   omega=omega, mean=mean.copy()))
        self.prev = np.concatenate([mean[:, 1:], mean[:, -1:]], axis=1)        # :92-93
        return mean[:, 0]                                                      # :94


class SPSA(OptimizerBase):
    """SPSAOptimizer  optimizers/spsa.py:48-117.  noise: {'rademacher': [iters][N,A,H,U] in {-1,+1}}."""

    def __init__(self, evaluator, low, high, horizon=50, max_iterations=5, population=500,
                 num_agents=5, alpha=0.602, gamma=0.101, a_par=0.01, noise_parameter=0.3):
        super().__init__(evaluator, low, high, horizon, num_agents, max_iterations)
        self.N = population
        self.alpha, self.gamma, self.a_par, self.c_par = F(alpha), F(gamma), F(a_par), F(noise_parameter)
        self.big_a = (F(max_iterations) / F(10.0)).astype(F)                   # :56
        self.params = self._init_mean()

    def reset(self):                                               # :119-127
        self.params = self._init_mean()

    def _optimize(self, state, noise):
        sol = self.params.copy()
        self.trace = []
        for it in range(self.iters):
            tf_ = F(it)
            ak = (self.a_par / pow32(((tf_ + F(1)).astype(F) + self.big_a).astype(F), self.alpha)).astype(F)  # :69
            ck = (self.c_par / pow32((tf_ + F(1)).astype(F), self.gamma)).astype(F)                             # :70
            delta = f32(noise["rademacher"][it])                               # :73-75
            step = (ck * delta).astype(F)
            pp = (sol + step).astype(F)                                        # :76
            pm = (sol - step).astype(F)                                        # :77
            ppf, pmf = self._clip_h(pp), self._clip_h(pm)                      # :78-81
            pen_p, pen_m = self._penalty(pp, ppf), self._penalty(pm, pmf)      # :82-89
            full = self.ev(state, np.concatenate([ppf, pmf], axis=0))          # :93-96
            rp = (full[:self.N] - pen_p).astype(F)                             # :98
            rm = (full[self.N:] - pen_m).astype(F)                             # :99
            den = ((F(2.0) * ck).astype(F) * delta).astype(F)
            g = ((rp - rm).astype(F)[:, :, None, None] / den).astype(F)
            ghat = (g.sum(axis=0, dtype=F) / F(self.N)).astype(F)              # :101-103
            sol = self._clip_h((sol + (ak * ghat).astype(F)).astype(F))        # :105-107
            self.trace.append(dict(rewards_plus=rp, rewards_minus=rm, ghat=ghat, solution=sol.copy(),
                                   ak=ak, ck=ck))
        self.params = np.concatenate([sol[:, 1:], sol[:, -1:]], axis=1)        # :114-115
        return sol[:, 0]


class PSO(OptimizerBase):
    """PSOOptimizer  optimizers/pso.py:47-160.
    noise: {'normal2': [iters][2] scalar N(0,1) (quirk Q3: shared by everything),
            'trunc': [N,A,H,U], 'uniform': [N,A,H,U]  (post-loop swarm re-seed :116-131)}
    reset noise: {'uniform_pos': [N,A,H,U], 'uniform_vel': [N,A,H,U]}.
    Quirk Q4: the constructor leaves pos/vel/pbest/pbest_r at ZERO (:50-59);
    only reset() randomises them and sets pbest_r = -inf."""

    def __init__(self, evaluator, low, high, horizon=50, max_iterations=5, population=500,
                 num_agents=5, c1=0.3, c2=0.5, w=0.2, initial_velocity_fraction=0.01):
        super().__init__(evaluator, low, high, horizon, num_agents, max_iterations)
        self.N = population
        self.c1, self.c2, self.w, self.v0f = F(c1), F(c2), F(w), F(initial_velocity_fraction)
        shp = (self.N, self.A, self.H, self.U)
        self.pos = np.zeros(shp, F)
        self.vel = np.zeros(shp, F)
        self.pbest = np.zeros(shp, F)
        self.pbest_r = np.zeros((self.N, self.A), F)
        self.gbest = np.zeros(shp[1:], F)
        self.gbest_r = np.zeros((self.A,), F)
        self.var = self._init_var()

    def _uniform(self, u01, lo, hi):
        return ((f32(u01) * (hi - lo).astype(F)).astype(F) + lo).astype(F)

    def reset(self, noise):                                        # :143-160
        self.pos = self._uniform(noise["uniform_pos"], self.lo_h, self.hi_h)
        v0 = (self.v0f * (self.hi_h - self.lo_h).astype(F)).astype(F)
        self.vel = self._uniform(noise["uniform_vel"], -v0, v0)
        self.pbest = self.pos.copy()
        self.pbest_r = np.full((self.N, self.A), -np.inf, F)
        self.gbest_r = np.full((self.A,), -np.inf, F)

    def _optimize(self, state, noise, rewards_override=None):
        """`rewards_override`: optional callable (iteration, rewards[N,A]) -> rewards[N,A] used by full-size parity
        tests for lock-step: it checks the oracle's rewards against the device's within tolerance and returns the
        device's, so that the exact comparisons below (pbest update, argmax) see identical values."""
        self.trace = []
        ar = np.arange(self.A)
        for it in range(self.iters):
            feas = self._clip_h(self.pos)                                      # :76-77
            pen = self._penalty(self.pos, feas)                                # :78-79
            self.pos = feas                                                    # :80
            rewards = (self.ev(state, self.pos) - pen).astype(F)               # :82
            if rewards_override is not None:
                rewards = f32(rewards_override(it, rewards))
            cond = self.pbest_r < rewards                                      # :84
            self.pbest = np.where(cond[:, :, None, None], self.pos, self.pbest)    # :86-88
            self.pbest_r = np.where(cond, rewards, self.pbest_r).astype(F)     # :89-91
            gi = argmax_first(self.pbest_r, axis=0)                            # :94
            self.gbest = self.pbest[gi, ar]                                    # :95-98
            # :99-103 as executed (quirk Q10): the reward is gathered from pbest_r [N,A] flattened N-major with the
            # index gi[a] + a*N that was built for the [A,N]-major position tensor -- the intended pbest_r[gi[a], a]
            # only when A == 1.  The Variable is re-filled with -inf after the loop (:137-138) and never read.
            self.gbest_r = self.pbest_r.reshape(-1)[gi + ar * self.N]
            r1, r2 = F(noise["normal2"][it][0]), F(noise["normal2"][it][1])
            t1 = (self.vel * self.w).astype(F)                                 # :104
            t2 = (((self.pbest - self.pos).astype(F) * self.c1).astype(F) * r1).astype(F)          # :105
            t3 = (((self.gbest[None] - self.pos).astype(F) * self.c2).astype(F) * r2).astype(F)    # :106
            self.vel = ((t1 + t2).astype(F) + t3).astype(F)
            self.pos = (self.pos + self.vel).astype(F)                         # :108
            self.trace.append(dict(rewards=rewards, gbest=self.gbest.copy(), gbest_r=self.gbest_r.copy(),
                                   gbest_idx=gi))
        solution = self.gbest[:, 0, :].copy()                                  # :114
        lb = (self.gbest - self.lo_h).astype(F)                                # :116
        ub = (self.hi_h - self.gbest).astype(F)                                # :117
        cv = np.minimum(np.minimum(((lb / F(2)) ** 2).astype(F), ((ub / F(2)) ** 2).astype(F)), self.var)
        shifted = np.concatenate([self.gbest[:, 1:], self.gbest[:, -1:]], axis=1)   # :123-125
        self.pos = ((f32(noise["trunc"]) * sqrt32(cv)).astype(F) + shifted).astype(F)   # :121-127
        v0 = (self.v0f * (self.hi_h - self.lo_h).astype(F)).astype(F)          # :128-129
        self.vel = self._uniform(noise["uniform"], -v0, v0)                    # :130-131
        self.pbest = self.pos.copy()                                           # :134
        self.pbest_r = np.full((self.N, self.A), -np.inf, F)                   # :135-136
        self.gbest_r = np.full((self.A,), -np.inf, F)                          # :137-138
        return solution


def cmaes_constants(N, k, n, alpha_cov=2.0):
    """CMAESOptimizer.__init__ constants  optimizers/cma_es.py:62-92,118-126 (fp32)."""
    kf = F(k)
    w = (log32((kf + F(0.5)).astype(F)) - log32(np.arange(1, k + 1).astype(F))).astype(F)   # :62-66
    w = np.concatenate([w, np.zeros((N - k,), F)])
    w = (w / w.sum(dtype=F)).astype(F)                                                      # :68
    mu_eff = ((w.sum(dtype=F) ** 2).astype(F) / (w * w).astype(F).sum(dtype=F)).astype(F)   # :69-70
    nf = F(n)
    c_sigma = ((mu_eff + F(2)) / ((nf + mu_eff).astype(F) + F(5))).astype(F)                # :73-75
    d_sigma = ((F(1) + F(2) * np.maximum(F(0), (sqrt32((mu_eff - F(1)) / (nf + F(1))) - F(1)).astype(F)))
               .astype(F) + c_sigma).astype(F)                                              # :76-79
    cc = ((F(4) + mu_eff / nf) / ((nf + F(4)).astype(F) + (F(2) * mu_eff / nf).astype(F))).astype(F)   # :81-83
    ac = F(alpha_cov)
    c1 = (ac / (((nf + F(1.3)) ** 2).astype(F) + mu_eff)).astype(F)                         # :86-88
    c_mu2 = (ac * ((mu_eff - F(2)).astype(F) + (F(1) / mu_eff).astype(F)).astype(F)
             / (((nf + F(2)) ** 2).astype(F) + (ac * mu_eff / F(2)).astype(F))).astype(F)   # :89-91
    c_mu = np.minimum((F(1) - c1).astype(F), c_mu2).astype(F)                               # :92
    e_norm = sqrt32((nf * ((F(1) - (F(1) / (F(4) * nf)).astype(F)).astype(F)
                           + (F(1) / (F(21) * (nf ** 2).astype(F))).astype(F)).astype(F)).astype(F))  # :118-126
    return dict(weights=w, mu_eff=mu_eff, c_sigma=c_sigma, d_sigma=d_sigma, cc=cc, c1=c1, c_mu=c_mu,
                e_norm=e_norm)


class CMAES(OptimizerBase):
    """CMAESOptimizer  optimizers/cma_es.py:43-227 (coupled-agents form, n = A*H*U).
    noise: {'normal': [iters][N, n]}.
    Quirks: Q5 y = z @ (B @ D) (:140);  Q6 rewards summed over agents (:158);
    h_sigma is a constant; state (m, sigma, C, B, D, p_sigma, p_C) persists
    across control steps and reset() restores only m and sigma (:215-227).
    The eigen-factorisation uses np.linalg.svd; its sign/order conventions are
    not TF's, so only iteration-0 samples and the deterministic update given
    identical samples are comparable."""

    def __init__(self, evaluator, low, high, horizon=50, max_iterations=5, population=500,
                 num_elite=50, num_agents=5, alpha_cov=2.0, h_sigma=1.0):
        super().__init__(evaluator, low, high, horizon, num_agents, max_iterations)
        self.N, self.k = population, num_elite
        self.n = self.A * self.H * self.U
        self.c = cmaes_constants(self.N, self.k, self.n, alpha_cov)
        self.h_sigma = F(h_sigma)
        self.m = self._init_mean().reshape(-1)
        self.sigma = sqrt32(self._init_var().reshape(-1))
        n = self.n
        self.C = np.eye(n, dtype=F)
        self.B = np.eye(n, dtype=F)
        self.D = np.eye(n, dtype=F)
        self.p_sigma = np.zeros((n,), F)
        self.p_C = np.zeros((n,), F)

    def reset(self):
        self.m = self._init_mean().reshape(-1)
        self.sigma = sqrt32(self._init_var().reshape(-1))

    @staticmethod
    def _mm(a, b):
        return (a.astype(np.float64) @ b.astype(np.float64)).astype(F)

    def _optimize(self, state, noise, eig=None, forced_order=None):
        c = self.c
        w = c["weights"]
        self.trace = []
        for it in range(self.iters):
            z = f32(noise["normal"][it])                                        # :139
            y = self._mm(z, self._mm(self.B, self.D))                           # :140
            samples = (self.m + (self.sigma * y).astype(F)).astype(F)           # :141
            samples = samples.reshape(self.N, self.A, self.H, self.U)           # :142
            feas = self._clip_h(samples)                                        # :147-148
            pen = self._penalty(samples, feas)                                  # :149-151
            rewards = (self.ev(state, feas) - pen).astype(F)                    # :157
            rsum = seq_sum(rewards, axis=1)                                     # :158
            order = topk_desc(rsum, self.N)                                     # :159
            if forced_order is not None:      # parity tests: keep lock-step across near-tied ranks
                order = np.asarray(forced_order(it, rsum, order))
            xs = feas[order].reshape(len(order), self.n)
            x_diff = (xs - self.m).astype(F)                                    # :161
            x_mean = (x_diff.astype(np.float64) * w[:len(order), None].astype(np.float64)).sum(0).astype(F)   # :162
            m = (self.m + x_mean).astype(F)                                     # :163
            y_mean = (x_mean / self.sigma).astype(F)                            # :167
            d_inv = np.diag((F(1) / np.diag(self.D)).astype(F)).astype(F)       # :168
            c_inv_half = self._mm(self._mm(self.B, d_inv), self.B.T)            # :169
            cs = c["c_sigma"]
            coef = sqrt32(((cs * (F(2) - cs).astype(F)).astype(F) * c["mu_eff"]).astype(F))
            p_sigma = (((F(1) - cs).astype(F) * self.p_sigma).astype(F)
                       + (coef * self._mm(c_inv_half, y_mean[:, None])[:, 0]).astype(F)).astype(F)   # :170-171
            nrm = sqrt32((p_sigma.astype(np.float64) ** 2).sum()).astype(F)
            sigma = (self.sigma * exp32(((cs / c["d_sigma"]).astype(F)
                                         * ((nrm / c["e_norm"]).astype(F) - F(1)).astype(F)).astype(F))).astype(F)  # :172-173
            cc = c["cc"]
            coef_c = (self.h_sigma * sqrt32(((cc * (F(2) - cc).astype(F)).astype(F) * c["mu_eff"]).astype(F))).astype(F)
            p_C = (((F(1) - cc).astype(F) * self.p_C).astype(F) + (coef_c * y_mean).astype(F)).astype(F)   # :177
            y_unw = (x_diff / self.sigma).astype(F)                             # :180
            # :181-182  map_fn(e * e^T) in fp32, times the weight in fp32, reduce_sum over the population.  TF's reduction
            # order is unspecified, so the SUM (here, in x_mean and in every matmul of this class) is accumulated in
            # float64 and rounded once -- the neutral reference the engine's fp32 orders are held to (the tolerances in
            # tests/test_gpu_cmaes.py, 2e-5 absolute, cover both); the PRODUCTS keep the reference's association
            yk = y_unw[:self.k]
            outer = (yk[:, :, None] * yk[:, None, :]).astype(F)                 # zero weights dropped
            y_s = (outer * w[:self.k, None, None]).astype(F).astype(np.float64).sum(0).astype(F)
            C = ((((F(1) - c["c1"]).astype(F) - c["c_mu"]).astype(F) * self.C).astype(F)
                 + ((c["c1"] * p_C[:, None]).astype(F) * p_C[None, :]).astype(F)).astype(F)   # :183  (c1 * p_C) * p_C^T
            C = (C + (c["c_mu"] * y_s).astype(F)).astype(F)                      # :183-184
            up = np.triu(C)                                                     # :188
            C = (up + np.triu(C, 1).T).astype(F)                                # :189-190
            if eig is None:
                u_, s_, _ = np.linalg.svd(C.astype(np.float64))                 # :195  s,U,_ = svd(C)
                s_, u_ = s_.astype(F), u_.astype(F)
            else:
                s_, u_ = eig[it]
            self.p_C, self.p_sigma, self.C, self.sigma = p_C, p_sigma, C, sigma     # :200-203
            self.B, self.D, self.m = u_, np.diag(sqrt32(s_)).astype(F), m           # :204-206
            self.trace.append(dict(samples=feas, rewards=rewards, order=order, m=m.copy(), sigma=sigma.copy(),
                                   p_sigma=p_sigma.copy(), p_C=p_C.copy(), C=C.copy(), s=s_.copy()))
        return self.m.reshape(self.A, self.H, self.U)[:, 0]                     # :211-212


# ---------------------------------------------------------------------------
# MPCPolicy.act  policies/mpc_policy.py:124-172
# ---------------------------------------------------------------------------
def policy_act(optimizer, observations, noise, exploration_noise=None):
    obs = np.asarray(observations)
    batched = obs
    if obs.ndim == 1:                                               # :150-152
        batched = np.tile(obs[None], (optimizer.A, 1))
    act, nxt, rew = optimizer.call(batched.astype(F), noise, exploration_noise)
    if obs.ndim == 1:                                               # :168-171
        return act[0], nxt[0], rew[0]
    return act, nxt, rew


# ---------------------------------------------------------------------------
# Synthetic problem builders shared by tests / bench
# ---------------------------------------------------------------------------
def make_mlp_params(dims, seed=42, last_scale=0.1):
    """Glorot-uniform kernels / zero biases (Keras Dense defaults,
    deterministic_mlp.py:21-24), last layer scaled so long rollouts stay finite."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(len(dims) - 1):
        lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
        w = rng.uniform(-lim, lim, size=(dims[i], dims[i + 1])).astype(F)
        if i == len(dims) - 2:
            w = (w * F(last_scale)).astype(F)
        ws.append(w)
        bs.append(np.zeros((dims[i + 1],), F))
    return ws, bs


def pendulum_start_states(num_agents, agent_offset=0):
    out = np.zeros((num_agents, 3), F)
    for a in range(num_agents):
        rng = np.random.default_rng(1234 + agent_offset + a)
        th = rng.uniform(-np.pi, np.pi)
        thd = rng.uniform(-1.0, 1.0)
        out[a] = [np.cos(th), np.sin(th), thd]
    return out


def cheetah_start_states(num_agents, dim_s=20, agent_offset=0):
    out = np.zeros((num_agents, dim_s), F)
    for a in range(num_agents):
        rng = np.random.default_rng(7 + agent_offset + a)
        out[a] = (rng.standard_normal(dim_s) * 0.1).astype(F)
    return out


def truncated_normal_noise(rng, shape):
    """Unit normal resampled until |z| < 2 (tf.random.truncated_normal semantics)."""
    z = rng.standard_normal(shape)
    bad = np.abs(z) >= 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) >= 2.0
    return z.astype(F)
