#!/usr/bin/env python
"""Device-resident control-step time of the persistent pendulum kernel (PI2 / CEM) against population size and
agent count: how much one agent's control step would gain from spreading its population over several CUs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rate(eng, A, steps=300):
    import torch
    from blackbox_mpc_amd.utils import synthetic as SY
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(SY.pendulum_start_states(A)).to(dev)
    nx = torch.empty_like(st)
    rec = torch.zeros((A, 5), device=dev)
    for _ in range(30):
        eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
        st, nx = nx, st
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
        st, nx = nx, st
    eng.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    for opt, name in ((L.OPT_PI2, "PI2"), (L.OPT_CEM, "CEM")):
        for A in (1, 8, 64):
            row = []
            for N in (250, 500, 1000, 2000):
                eng = Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=30,
                             population_size=N, max_iterations=5, num_elite=50, lamda=1.0, seed=0)
                row.append("N=%d %.1f us" % (N, rate(eng, A)))
            print(name, "A=%d" % A, " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
