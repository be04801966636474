"""Generates tests/golden/*.npz from the oracle (oracle/oracle_np.py) on seeded inputs.

The reference ships no golden vectors and cannot be executed here (TensorFlow 2.0 absent), so these fixtures
are ORACLE outputs, committed to (a) freeze the oracle against accidental change (CPU test) and (b) give the
GPU tests a fixed set of expected values that does not depend on the oracle code that runs on the GPU box.
Shrunken versions of BASELINE configs 1-5, plus cfg6 (SPSA) and cfg7 (CMA-ES iteration 0).  Run from the repo root:
python tests/golden/make_golden.py [cfgN ...]   (no argument: rewrite every fixture)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as O  # noqa: E402

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))


def pend_eval():
    return O.Evaluator("pendulum", O.Handler(O.pendulum_dynamics, True))


def cheetah_problem(seed=42):
    S, U = 20, 6
    ws, bs = O.make_mlp_params([26, 200, 200, 20], seed=seed)
    rng = np.random.default_rng(seed + 1)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    st = np.random.default_rng(seed + 2)
    stats = [st.normal(0, 0.2, S).astype(F), st.uniform(0.5, 1.5, S).astype(F), st.normal(0, 0.1, U).astype(F),
             st.uniform(0.5, 1.5, U).astype(F), st.normal(0, 0.01, S).astype(F), st.uniform(0.05, 0.15, S).astype(F)]
    ev = O.Evaluator("cheetah", O.Handler(O.MLP(ws, bs, ["tanh", "tanh", None]), False, True, stats))
    return ws, bs, stats, ev


def main():
    out = {}
    # cfg1-like: RandomSearch, pendulum
    rng = np.random.default_rng(101)
    N, A, H = 200, 1, 20
    st = O.pendulum_start_states(A)
    u01 = rng.random((N, A, H, 1)).astype(F)
    rs = O.RandomSearch(pend_eval(), [-2.0], [2.0], horizon=H, population=N, num_agents=A)
    a, n, r = rs.call(st, {"uniform": u01})
    out["cfg1"] = dict(states=st, uniform=u01, rewards=rs.trace[0]["rewards"], best=rs.trace[0]["best"], action=a,
                       next_state=n, reward=r)
    # cfg2-like: CEM pendulum N=500 H=30 (full size), 3 iterations
    rng = np.random.default_rng(102)
    N, A, H, iters, k = 500, 1, 30, 3, 50
    st = O.pendulum_start_states(A)
    xi = np.stack([O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)])
    cem = O.CEM(pend_eval(), [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_elite=k, num_agents=A)
    a, n, r = cem.call(st, {"trunc": list(xi)})
    out["cfg2"] = dict(states=st, trunc=xi, rewards=np.stack([t["rewards"] for t in cem.trace]),
                       elites=np.stack([t["elites"] for t in cem.trace]), mean=np.stack([t["mean"] for t in cem.trace]),
                       var=np.stack([t["var"] for t in cem.trace]), action=a, next_state=n, reward=r)
    # cfg3-like: PI2 pendulum, 4 agents, 2 control steps (warm start)
    rng = np.random.default_rng(103)
    N, A, H, iters = 256, 4, 30, 2
    st = O.pendulum_start_states(A)
    pi2 = O.PI2(pend_eval(), [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_agents=A, lamda=1.0)
    xs, acts, prevs = [], [], []
    for step in range(2):
        xi = np.stack([O.truncated_normal_noise(rng, (N, A, H, 1)) for _ in range(iters)])
        a, n, r = pi2.call(st, {"trunc": list(xi)})
        xs.append(xi); acts.append(a); prevs.append(pi2.prev.copy())
    out["cfg3"] = dict(states=st, trunc=np.stack(xs), action=np.stack(acts), prev_mean=np.stack(prevs))
    # cfg4-like: evaluator on the learned MLP (cheetah), 64 particles x 30 steps + one model step
    rng = np.random.default_rng(104)
    ws, bs, stats, ev = cheetah_problem()
    N, A, H = 64, 2, 30
    st = O.cheetah_start_states(A, 20)
    seq = rng.uniform(-1, 1, (N, A, H, 6)).astype(F)
    s1 = rng.normal(0, 0.5, (32, 20)).astype(F)
    a1 = rng.uniform(-1, 1, (32, 6)).astype(F)
    out["cfg4"] = dict(states=st, seq=seq, rewards=ev(st, seq), step_states=s1, step_actions=a1,
                       step_next=ev.predict_next_state(s1, a1), mlp_seed=np.array(42))
    # cfg5-like: PSO on pendulum after reset (2 agents) -- swarm state after one control step
    rng = np.random.default_rng(105)
    N, A, H, iters = 96, 2, 8, 3
    st = O.pendulum_start_states(A)
    pso = O.PSO(pend_eval(), [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_agents=A)
    rn = {"uniform_pos": rng.random((N, A, H, 1)).astype(F), "uniform_vel": rng.random((N, A, H, 1)).astype(F)}
    pso.reset(rn)
    noise = {"normal2": rng.standard_normal((iters, 2)).astype(F), "trunc": O.truncated_normal_noise(rng, (N, A, H, 1)),
             "uniform": rng.random((N, A, H, 1)).astype(F)}
    a, n, r = pso.call(st, noise)
    out["cfg5"] = dict(states=st, reset_pos=rn["uniform_pos"], reset_vel=rn["uniform_vel"], normal2=noise["normal2"],
                       trunc=noise["trunc"], uniform=noise["uniform"], action=a, pos=pso.pos, vel=pso.vel,
                       gbest=pso.gbest, rewards=np.stack([t["rewards"] for t in pso.trace]))
    # cfg6: SPSA on pendulum (spsa.py:61-117), 3 agents, two control steps (shift-left warm start :114-115)
    rng = np.random.default_rng(106)
    N, A, H, iters = 64, 3, 10, 3
    st = O.pendulum_start_states(A)
    spsa = O.SPSA(pend_eval(), [-2.0], [2.0], horizon=H, max_iterations=iters, population=N, num_agents=A)
    rads, acts, params, ghats = [], [], [], []
    for step in range(2):
        rad = (rng.integers(0, 2, (iters, N, A, H, 1)) * 2 - 1).astype(F)
        a, n, r = spsa.call(st, {"rademacher": list(rad)})
        rads.append(rad); acts.append(a); params.append(spsa.params.copy())
        ghats.append(np.stack([t["ghat"] for t in spsa.trace]))
    out["cfg6"] = dict(states=st, rademacher=np.stack(rads), action=np.stack(acts), params=np.stack(params),
                       ghat=np.stack(ghats))
    # cfg7: CMA-ES on pendulum (cma_es.py:129-213), coupled agents, ONE iteration from the constructor state (B = D = I,
    # so the samples do not depend on anybody's SVD conventions) + the deterministic (m, p_sigma, sigma, p_C, C) update;
    # D = sqrt(singular values of C), descending, is convention-free as well
    rng = np.random.default_rng(107)
    N, A, H, k = 96, 2, 6, 12
    n = A * H * 1
    st = O.pendulum_start_states(A)
    z = rng.standard_normal((1, N, n)).astype(F)
    cma = O.CMAES(pend_eval(), [-2.0], [2.0], horizon=H, max_iterations=1, population=N, num_elite=k, num_agents=A)
    a, nx, r = cma.call(st, {"normal": list(z)})
    tr = cma.trace[0]
    out["cfg7"] = dict(states=st, normal=z, action=a, next_state=nx, reward=r, samples=tr["samples"], rewards=tr["rewards"],
                       order=tr["order"][:k].astype(np.int32), m=tr["m"], sigma=tr["sigma"], p_sigma=tr["p_sigma"], p_C=tr["p_C"],
                       C=tr["C"], D=np.diag(cma.D).copy())
    only = set(sys.argv[1:])
    for name, d in out.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k_: np.asarray(v).shape for k_, v in d.items()})


if __name__ == "__main__":
    main()
