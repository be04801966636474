#!/usr/bin/env python
"""Bisects a parity failure of the quad rollout kernel (run on a GPU box): evaluates N x H candidate sequences with
parts of the network zeroed out (hidden features 192..199 of layer 0 / of layer 1) and prints the error against the C
oracle for each variant and horizon."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    from oracle import oracle_c as OC
    S, U, N, A = 20, 6, 64, 1
    rng = np.random.default_rng(3)
    ws0, bs0 = SY.make_mlp_params()
    bs0 = [rng.normal(0, 0.05, b.shape).astype(F) for b in bs0]
    stats = [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F), rng.normal(0, 0.1, U).astype(F),
             rng.uniform(0.5, 1.5, U).astype(F), rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]
    lo, hi = [-1.0] * U, [1.0] * U
    states = SY.cheetah_start_states(A)
    for name in ("full", "h0[192:]=0 (W1 rows zero)", "h1[192:]=0 (W2 rows zero)", "both", "only h0[192:] (W1 rows < 192 zero)",
                 "only h1[192:] (W2 rows < 192 zero)"):
        ws = [w.copy() for w in ws0]
        bs = [b.copy() for b in bs0]
        if name.startswith("h0") or name == "both":
            ws[1][192:, :] = 0
        if name.startswith("h1") or name == "both":
            ws[2][192:, :] = 0
        if name.startswith("only h0"):
            ws[1][:192, :] = 0
        if name.startswith("only h1"):
            ws[2][:192, :] = 0
        for H in (1, 2, 5):
            seq = rng.uniform(-1, 1, (N, A, H, U)).astype(F)
            co = OC.COracle("mlp", "cheetah", lo, hi, N, A, H, S, mlp=(ws, bs, ["tanh", "tanh", None]), stats=stats)
            want = co.evaluate(states, seq)
            ev = Engine(L.OPT_NONE, L.DYN_MLP, L.REW_CHEETAH, lo, hi, dim_s=S, num_agents=A, planning_horizon=H,
                        population_size=0, max_iterations=0, num_elite=0)
            ev.set_mlp(ws, bs, [1, 1, 0], stats)
            got = ev.evaluate(states, seq)
            print("%-40s H=%d  max err %.3g  (max |want| %.3g)" % (name, H, float(np.abs(got - want).max()), float(np.abs(want).max())), flush=True)


if __name__ == "__main__":
    main()
