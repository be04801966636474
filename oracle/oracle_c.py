"""ctypes loader for the C restatement (oracle/oracle_c.c) -- TEST INFRASTRUCTURE ONLY.

Same rules as oracle_np.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
MAX_LAYERS = 8
DYN = {"pendulum": 1, "mlp": 2}
REW = {"pendulum": 1, "cheetah": 2}
ACT = {None: 0, "tanh": 1, "relu": 2, "sigmoid": 3}
OPT = {"RandomSearch": 1, "CEM": 2, "PI2": 3}
_fp = ctypes.POINTER(ctypes.c_float)


class Problem(ctypes.Structure):
    _fields_ = [("dyn", ctypes.c_int32), ("rew", ctypes.c_int32), ("as_executed", ctypes.c_int32),
                ("N", ctypes.c_int32), ("A", ctypes.c_int32), ("H", ctypes.c_int32), ("U", ctypes.c_int32),
                ("S", ctypes.c_int32), ("iters", ctypes.c_int32), ("k", ctypes.c_int32),
                ("alpha", ctypes.c_float), ("lamda", ctypes.c_float),
                ("lo", _fp), ("hi", _fp),
                ("n_layers", ctypes.c_int32), ("normalized", ctypes.c_int32),
                ("dims", ctypes.c_int32 * (MAX_LAYERS + 1)), ("acts", ctypes.c_int32 * MAX_LAYERS),
                ("w", _fp * MAX_LAYERS), ("b", _fp * MAX_LAYERS),
                ("mean_s", _fp), ("std_s", _fp), ("mean_a", _fp), ("std_a", _fp), ("mean_t", _fp), ("std_t", _fp)]


def build(force=False):
    """Compile oracle_c.c (gcc, OpenMP) if the shared object is missing or older than the source."""
    src = os.path.join(_HERE, "oracle_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.bbo_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(_fp)


class COracle:
    """One problem instance: dynamics + reward + optimizer sizes.  Mirrors oracle_np's Evaluator / optimizers."""

    def __init__(self, dynamics, reward, low, high, N, A, H, S, iters=1, k=1, alpha=0.25, lamda=1.0,
                 as_executed=True, mlp=None, stats=None):
        self.lo = np.ascontiguousarray(low, np.float32).reshape(-1)
        self.hi = np.ascontiguousarray(high, np.float32).reshape(-1)
        self.N, self.A, self.H, self.U, self.S, self.iters, self.k = N, A, H, self.lo.size, S, iters, k
        self._keep = [self.lo, self.hi]
        P = Problem(dyn=DYN[dynamics], rew=REW[reward], as_executed=int(as_executed), N=N, A=A, H=H, U=self.U, S=S,
                    iters=iters, k=k, alpha=alpha, lamda=lamda, lo=_p(self.lo), hi=_p(self.hi))
        if mlp is not None:
            weights, biases, acts = mlp
            P.n_layers = len(weights)
            for l, (w, b, a) in enumerate(zip(weights, biases, acts)):
                w = np.ascontiguousarray(w, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                self._keep += [w, b]
                P.dims[l], P.dims[l + 1] = w.shape
                P.acts[l] = ACT[a]
                P.w[l], P.b[l] = _p(w), _p(b)
            if stats is not None:
                P.normalized = 1
                st = [np.ascontiguousarray(v, np.float32) for v in stats]
                self._keep += st
                P.mean_s, P.std_s, P.mean_a, P.std_a, P.mean_t, P.std_t = [_p(v) for v in st]
        self.P = P
        self.prev_mean = self.init_mean()

    def init_mean(self):
        m = ((self.lo + self.hi).astype(np.float32) / np.float32(2)).astype(np.float32)
        return np.ascontiguousarray(np.tile(m, (self.A, self.H, 1)), np.float32)

    def reset(self):
        self.prev_mean = self.init_mean()

    def evaluate(self, states, seq):
        states = np.ascontiguousarray(states, np.float32)
        seq = np.ascontiguousarray(seq, np.float32)
        P = Problem.from_buffer_copy(self.P)
        P.N = seq.shape[0]
        out = np.empty((P.N, self.A), np.float32)
        rc = lib().bbo_evaluate(ctypes.byref(P), _p(states), _p(seq), _p(out))
        assert rc == 0
        return out

    def predict_next_state(self, s, a):
        s = np.ascontiguousarray(s, np.float32)
        a = np.ascontiguousarray(a, np.float32)
        out = np.empty_like(s)
        lib().bbo_predict_next_state(ctypes.byref(self.P), s.shape[0], _p(s), _p(a), _p(out))
        return out

    def evaluate_next_reward(self, cur, nxt, act):
        cur, nxt, act = [np.ascontiguousarray(v, np.float32) for v in (cur, nxt, act)]
        out = np.empty((cur.shape[0],), np.float32)
        lib().bbo_evaluate_next_reward(ctypes.byref(self.P), cur.shape[0], _p(cur), _p(nxt), _p(act), _p(out))
        return out

    def as_evaluator(self):
        """Adapter with oracle_np.Evaluator's interface, so that the NumPy optimizers (PSO, SPSA, CMA-ES) can run
        BASELINE-size problems with this library doing the rollouts."""
        co = self

        class _Ev:
            def __call__(self, states, seq, return_final_state=False):
                assert not return_final_state
                return co.evaluate(states, seq)

            def predict_next_state(self, s, a):
                return co.predict_next_state(s, a)

            def evaluate_next_reward(self, cur, nxt, act):
                return co.evaluate_next_reward(cur, nxt, act)
        return _Ev()

    def optimize(self, opt, state, noise=None, seed=0, forced_elites=None, trace=False):
        """One OptimizerBase.__call__ (exploration noise off).  noise: [iters][N,A,H,U] standard draws or None."""
        state = np.ascontiguousarray(state, np.float32)
        A, U, S, HU = self.A, self.U, self.S, self.H * self.U
        iters = 1 if opt == "RandomSearch" else self.iters
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(np.stack([np.asarray(x, np.float32) for x in noise]), np.float32)
            assert nz.shape == (iters, self.N, A, self.H, U)
        fe = None if forced_elites is None else np.ascontiguousarray(forced_elites, np.int32)
        action, nxt, rew = np.empty((A, U), np.float32), np.empty((A, S), np.float32), np.empty((A,), np.float32)
        el = np.empty((iters, A, self.k), np.int32) if trace else None
        mean, var = (np.empty((A, self.H, U), np.float32), np.empty((A, self.H, U), np.float32)) if trace else (None, None)
        ip = ctypes.POINTER(ctypes.c_int32)
        rc = lib().bbo_optimize(ctypes.byref(self.P), OPT[opt], _p(state), _p(nz) if nz is not None else None,
                                ctypes.c_uint64(seed), _p(self.prev_mean), _p(action), _p(nxt), _p(rew),
                                el.ctypes.data_as(ip) if trace else None, _p(mean) if trace else None,
                                _p(var) if trace else None, fe.ctypes.data_as(ip) if fe is not None else None)
        assert rc == 0
        if trace:
            return action, nxt, rew, dict(elites=el, mean=mean, var=var)
        return action, nxt, rew


def num_threads():
    return lib().bbo_num_threads()


def set_num_threads(n):
    lib().bbo_set_num_threads(int(n))
