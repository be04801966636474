#!/usr/bin/env python
"""Learned-dynamics Pendulum as the reference's model-based-RL tutorials set it up (tutorials/model_based_RL/
tutorial_one.py: DeterministicMLP 4-32-32-32-3, tanh x3 + linear, pendulum reward, CEM defaults): device-resident
control-step time and the kernel the engine picks."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def rate(eng, A, steps=100):
    import torch
    from blackbox_mpc_amd.utils import synthetic as SY
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(SY.pendulum_start_states(A)).to(dev)
    nx = torch.empty_like(st)
    rec = torch.zeros((A, 5), device=dev)
    for _ in range(10):
        eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
        st, nx = nx, st
    eng.synchronize()
    eng.set_profiling(True, 1)
    eng.get_profile()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
        st, nx = nx, st
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e6
    ms, n, name = eng.get_profile()
    return dt, name, ms * 1e3 / max(n, 1)


def main():
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    dims = [4, 32, 32, 32, 3]
    ws, bs = SY.make_mlp_params(dims, seed=3, last_scale=0.1)
    stats = [np.zeros(3, F), np.ones(3, F), np.zeros(1, F), np.ones(1, F), np.zeros(3, F), np.full(3, 0.1, F)]
    for opt, name in ((L.OPT_CEM, "CEM"), (L.OPT_PI2, "PI2")):
        for A, N, H in ((1, 500, 50), (5, 500, 50), (1, 500, 30), (1, 1000, 30), (10, 1000, 30)):
            eng = Engine(opt, L.DYN_MLP, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                         population_size=N, max_iterations=5, num_elite=50, lamda=1.0, seed=0)
            eng.set_mlp(ws, bs, [1, 1, 1, 0], stats)
            dt, kn, kus = rate(eng, A)
            print("%s A=%d N=%d H=%d: %.1f us/control step; %s %.1f us/launch" % (name, A, N, H, dt, kn, kus), flush=True)


if __name__ == "__main__":
    main()
