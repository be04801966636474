#!/usr/bin/env python
"""Kernel-iteration probe for the learned-dynamics rollout kernels (run on a GPU box).

For each variant named on the command line (VARIANT = "ENV=VAL[,ENV=VAL...]" or "default") it
  1. evaluates N x H candidate sequences through bbmpc_evaluate and compares with the C oracle (tests/ tolerances),
  2. runs a PI2 control-step loop (north-star target 2: N=1000, H=30, 5 iterations) with HIP events around the
     dominant kernel and prints its average launch time and the device-resident control-step time.
Usage: python tools/q4_probe.py [--n 1000] [--h 30] [--agents 1] default BBMPC_MLP_Q4V=1 ...
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--h", type=int, default=30)
    ap.add_argument("--agents", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--opt", default="pi2")
    ap.add_argument("variants", nargs="*", default=["default"])
    args = ap.parse_args()
    from blackbox_mpc_amd import _build
    _build.build()
    import torch
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    from oracle import oracle_c as OC
    S, U, N, H, A = 20, 6, args.n, args.h, args.agents
    ws, bs = SY.make_mlp_params()
    rng = np.random.default_rng(3)
    bs = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs]
    stats = [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F), rng.normal(0, 0.1, U).astype(F),
             rng.uniform(0.5, 1.5, U).astype(F), rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]
    lo, hi = [-1.0] * U, [1.0] * U
    states = SY.cheetah_start_states(A)
    seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
    co = OC.COracle("mlp", "cheetah", lo, hi, N, A, H, S, mlp=(ws, bs, ["tanh", "tanh", None]), stats=stats)
    want = co.evaluate(states, seq)
    dev = torch.device("cuda", 0)
    for var in args.variants:
        envs = {} if var == "default" else dict(kv.split("=", 1) for kv in var.split(","))
        for k, v in envs.items():
            os.environ[k] = v
        try:
            ev = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=A, planning_horizon=H,
                        population_size=0, max_iterations=0, num_elite=0)
            ev.set_mlp(ws, bs, [1, 1, 0], stats)
            got = ev.evaluate(states, seq)
            err = np.abs(got - want)
            tol = 1e-3 * np.abs(want) + 1e-3 * H
            ok = bool(np.all(err <= tol))
            opt = {"pi2": L.OPT_PI2, "cem": L.OPT_CEM}[args.opt]
            eng = Engine(opt, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=A, planning_horizon=H,
                         population_size=N, max_iterations=5, num_elite=50, lamda=1.0, seed=0)
            eng.set_mlp(ws, bs, [1, 1, 0], stats)
            st = torch.from_numpy(states).to(dev)
            nx = torch.empty_like(st)
            rec = torch.zeros((A, U + S + 1), device=dev)
            for _ in range(10):
                eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
            eng.synchronize()
            step_us = (time.perf_counter() - t0) / args.steps * 1e6
            eng.set_profiling(True, 1)
            for _ in range(8):
                eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
            eng.synchronize()
            ms, n, name = eng.get_profile()
            print("%-40s parity %s (max err %.3g, max |want| %.3g)  kernel %s %.2f us x %d   control step %.1f us" %
                  (var, "OK" if ok else "FAIL", float(err.max()), float(np.abs(want).max()), name, ms / max(n, 1) * 1e3, n, step_us),
                  flush=True)
        finally:
            for k in envs:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
