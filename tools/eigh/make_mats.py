"""Input file of tools/eigh/eigh_probe.bin: covariance matrices of config 5's CMA-ES leg (n = 300) taken from a closed
loop -- the rank-deficient ones of the first iterations (C = alpha I + low rank) and the full-rank ones later.  Run on
the GPU box; writes gpurun_out/eigh_mats.bin (int32 n, int32 count, count x n x n float32): copy it to _probe/.
    python tools/eigh/make_mats.py [control steps, default 24]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.engine import Engine
from blackbox_mpc_amd.utils import synthetic as SY

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    S, U, H, A, N, k, iters = 20, 6, 50, 1, 2000, 50, 5
    eng = Engine(L.OPT_CMAES, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=A, planning_horizon=H,
                 population_size=N, max_iterations=iters, num_elite=k, seed=0, quirks=L.CMAES_PER_AGENT)
    dims = [S + U, 200, 200, S]
    ws, bs = SY.make_mlp_params(dims, seed=42)
    eng.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], list(SY.cheetah_stats(S, U)))
    eng.set_trace(True)
    state = SY.cheetah_start_states(A, S)
    keep = {(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (1, 0), (1, 4), (2, 4), (4, 4), (8, 4), (15, 4), (steps - 1, 4)}
    mats = []
    for t in range(steps):
        act, state, rew = eng.optimize(state)
        for it in range(iters):
            if (t, it) in keep:
                mats.append(eng.get_trace(it, L.TRACE_CMA_C).astype(np.float32).reshape(-1, H * U, H * U)[0])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/eigh_mats.bin", "wb") as f:
        np.array([H * U, len(mats)], np.int32).tofile(f)
        np.stack(mats).astype(np.float32).tofile(f)
    ev = [np.linalg.eigvalsh(m.astype(np.float64)) for m in mats]
    for i, e in enumerate(ev):
        print("matrix %2d: eigenvalues %.5f .. %.5f" % (i, e[0], e[-1]))

if __name__ == "__main__":
    main()
