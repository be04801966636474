#!/usr/bin/env python
"""Host-in / host-out control step of one rank with and without the record all-gather behind it (one-rank RCCL
communicator on one GPU): Engine.optimize vs Engine.optimize_gather, median wall time per call (config 2)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(f, n=2000, w=100):
    for i in range(w):
        f(i)
    ts = np.empty(n)
    for i in range(n):
        t0 = time.perf_counter()
        f(w + i)
        ts[i] = time.perf_counter() - t0
    return np.median(ts) * 1e6, np.percentile(ts, 10) * 1e6, np.percentile(ts, 90) * 1e6


def main():
    import torch
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    dev = torch.device("cuda", 0)

    def make():
        return Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=30,
                      population_size=500, max_iterations=5, num_elite=50, seed=0)
    plain, gat = make(), make()
    gat.comm_init(Engine.comm_unique_id(), 1, 0)
    gathered = [torch.zeros((1, 5), device=dev) for _ in range(2)]
    st = {"p": SY.pendulum_start_states(1), "g": SY.pendulum_start_states(1)}

    def f_plain(i):
        _, st["p"], _ = plain.optimize(st["p"], i)

    def f_gather(i):
        b = i & 1
        gat.gather_wait(b)
        _, st["g"], _ = gat.optimize_gather(st["g"], gathered[b].data_ptr(), b, i)
    print("Engine.optimize                      median %.2f us  p10 %.2f  p90 %.2f" % med(f_plain))
    print("gather_wait + Engine.optimize_gather median %.2f us  p10 %.2f  p90 %.2f" % med(f_gather))


if __name__ == "__main__":
    main()
