"""Why the block Jacobi needs five and more sweeps on the CMA-ES covariance, in numbers.

Runs the config-5 CMA-ES shape (HalfCheetah MLP, N = 2000, H = 50, per-agent instances of n = 300) closed loop with the
BLOCK JACOBI (BBMPC_CMA_EIGH=0) and the parity trace on, and prints for every decomposition of every control step
  * the column pairs rotated per sweep (BBMPC_TRACE_CMA_SVD_STATS) and the number of sweeps,
  * how good a warm start the previous eigenvectors are:  |offdiag(B0^T C B0)|_F  against the width of C's spectrum
    (a perturbation small against the eigenvalue gaps would converge in one or two sweeps; here the off-diagonal part
    is LARGER than the whole spectrum is wide, because the rank-51 update of one iteration has Frobenius norm
    c_mu * 300 * sqrt(sum w_i^2) = 0.03 while the spectrum is ~0.01-0.1 wide).
usage: BBMPC_CMA_EIGH=0 python tools/cma_sweep_histogram.py [control_steps] > profiles/r4_cfg5cma_sweeps.md"""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("BBMPC_CMA_EIGH", "0")
from blackbox_mpc_amd import _lib as L  # noqa: E402
from blackbox_mpc_amd.engine import Engine  # noqa: E402
from blackbox_mpc_amd.utils import synthetic as SY  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S, U, H, N, A, k, iters = 20, 6, 50, 2000, 4, 50, 5
eng = Engine(L.OPT_CMAES, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H,
             population_size=N, max_iterations=iters, num_elite=k, seed=0, quirks=L.CMAES_PER_AGENT)
ws, bs = SY.make_mlp_params()
stats = SY.cheetah_stats(S, U)
eng.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
eng.set_trace(True)
state = SY.cheetah_start_states(A, S)
n = H * U
hist = collections.Counter()
rows = []
Bprev = np.tile(np.eye(n, dtype=np.float32), (A, 1, 1))
for step in range(steps):
    act, nxt, rew = eng.optimize(state)
    for it in range(iters):
        st = eng.get_trace(it, L.TRACE_CMA_SVD_STATS)
        C = eng.get_trace(it, L.TRACE_CMA_C).astype(np.float64)
        for g in range(A):
            rot = [int(x) for x in st[g, :15]]
            sweeps = max([i + 1 for i, x in enumerate(rot) if x] or [0]) + (1 if any(rot) else 0)   # + the clean verification sweep
            hist[sweeps] += 1
            M = Bprev[g].astype(np.float64).T @ C[g] @ Bprev[g]
            off = np.linalg.norm(M - np.diag(np.diag(M)))
            ev = np.linalg.eigvalsh(C[g])
            if g == 0:
                rows.append((step, it, sweeps, rot[:sweeps], off, ev[-1] - ev[0], np.median(np.diff(ev))))
        Bprev = eng.get_trace(it, L.TRACE_CMA_B)
    state = nxt
print("# Block Jacobi on the CMA-ES covariance (config-5 shape, n = 300): sweeps and rotations per decomposition\n")
print("`BBMPC_CMA_EIGH=0 python tools/cma_sweep_histogram.py %d` -- %d control steps x %d iterations x %d instances, closed loop from a fresh episode.\n" % (steps, steps, iters, A))
print("## sweeps per decomposition (including the final sweep that finds nothing to rotate)\n")
print("| sweeps | decompositions |\n|---|---|")
for s_ in sorted(hist):
    print("| %d | %d |" % (s_, hist[s_]))
print("\n## instance 0: rotated column pairs per sweep (of 44 850), warm start quality\n")
print("| control step | iteration | sweeps | rotations per sweep | off-diagonal norm of B0^T C B0 | spectrum width | median eigenvalue gap |\n|---|---|---|---|---|---|---|")
for r in rows:
    print("| %d | %d | %d | %s | %.3e | %.3e | %.1e |" % (r[0], r[1], r[2], " ".join(str(x) for x in r[3]), r[4], r[5], r[6]))
