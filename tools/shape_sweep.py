#!/usr/bin/env python
"""Which rollout kernel a learned network gets, and what it costs off the benchmark shape (run on a GPU box).

PI2 and CEM control steps (N = 1000, H = 30, 5 iterations unless --n / --h say otherwise) through networks other than the
26-200-200-20 tanh family the fast kernels are compiled for -- including the two the reference itself pins:
tutorials/mujoco/tutorial_two.py:23-33 (26-500-500-500-20) and tutorials/low_level_api/tutorial_one.py:38-84 (4-32-32-32-3).
Per shape: the dominant kernel, its average launch time (HIP events on the launch stream), the fraction of the fp32 matrix
peak that is (2 * sum(in*out) FLOP per particle and model step), the device-resident control step, and "x family": time per
FLOP relative to the family kernel at the same population.  Prints a markdown table (profiles/r6_shape_sweep.md).

    python tools/shape_sweep.py [--n 1000] [--h 30] [--steps 40]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32
PEAK = 157.3e12

# a HalfCheetah-style reward for observation vectors shorter than 18 (the built-in indexes state[17], cost_func.py:18):
# same terms on the forward velocity stock HalfCheetah-v2 keeps at index 8
REWARD_S17 = """
__device__ float bbmpc_user_reward(const float* cur, const float* act, const float* nxt, int S, int U) {
    float r = 0.0f;
    if (cur[5] >= 0.2f) r = r + (-10.0f);
    if (cur[6] >= 0.0f) r = r + (-10.0f);
    if (cur[7] >= 0.0f) r = r + (-10.0f);
    r = r + nxt[8];
    float ss = 0.0f;
    for (int u = 0; u < U; ++u) ss = ss + act[u] * act[u];
    return r - 0.1f * ss;
}
"""

SHAPES = [
    # label, dims, activations, S, U, reward
    ("26-200-200-20 tanh (the family)", [26, 200, 200, 20], ["tanh", "tanh", None], 20, 6, "cheetah"),
    ("23-200-200-17 tanh (stock HalfCheetah-v2)", [23, 200, 200, 17], ["tanh", "tanh", None], 17, 6, "user17"),
    ("26-256-256-20 tanh", [26, 256, 256, 20], ["tanh", "tanh", None], 20, 6, "cheetah"),
    ("26-200-200-20 relu", [26, 200, 200, 20], ["relu", "relu", None], 20, 6, "cheetah"),
    ("26-500-500-500-20 (mujoco/tutorial_two.py:23-33)", [26, 500, 500, 500, 20], ["tanh", "tanh", "tanh", None], 20, 6, "cheetah"),
    ("4-32-32-32-3 (low_level_api/tutorial_one.py:38-84)", [4, 32, 32, 32, 3], ["tanh", "tanh", "tanh", None], 3, 1, "pendulum"),
    ("26-32-32-32-20 (mujoco/tutorial_one.py:22-31)", [26, 32, 32, 32, 20], ["tanh", "tanh", "tanh", None], 20, 6, "cheetah"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--h", type=int, default=30)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--opts", default="PI2,CEM")
    args = ap.parse_args()
    from blackbox_mpc_amd import _build
    _build.build()
    import torch
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    ACT = {None: L.ACT_NONE, "tanh": L.ACT_TANH, "relu": L.ACT_RELU, "sigmoid": L.ACT_SIGMOID}
    dev = torch.device("cuda", 0)
    N, H, A = args.n, args.h, 1
    print("| network | optimizer | kernel | us per launch | of fp32 matrix peak | control step us | x family (time per FLOP) |")
    print("|---|---|---|---|---|---|---|")
    family = {}
    for label, dims, acts, S, U, reward in SHAPES:
        ws, bs = SY.make_mlp_params(dims, seed=42)
        rng = np.random.default_rng(7)
        stats = [rng.normal(0, 0.2, S).astype(F), rng.uniform(0.5, 1.5, S).astype(F), rng.normal(0, 0.1, U).astype(F),
                 rng.uniform(0.5, 1.5, U).astype(F), rng.normal(0, 0.01, S).astype(F), rng.uniform(0.05, 0.15, S).astype(F)]
        flop = 2.0 * sum(a * b for a, b in zip(dims[:-1], dims[1:])) * N * A * H
        lo, hi = ([-2.0], [2.0]) if U == 1 else ([-1.0] * U, [1.0] * U)
        rk = {"cheetah": L.REW_CHEETAH, "pendulum": L.REW_PENDULUM, "user17": L.REW_USER}[reward]
        for opt_name in args.opts.split(","):
            opt = {"PI2": L.OPT_PI2, "CEM": L.OPT_CEM}[opt_name]
            eng = Engine(opt, L.DYN_MLP, rk, lo, hi, dim_s=S, num_agents=A, planning_horizon=H, population_size=N,
                         max_iterations=5, num_elite=50, lamda=1.0, seed=0)
            if reward == "user17":
                eng.set_reward_source(REWARD_S17)
            eng.set_mlp(ws, bs, [ACT[a] for a in acts], stats)
            st = torch.from_numpy(SY.pendulum_start_states(A) if S == 3 else rng.normal(0, 0.3, (A, S)).astype(F)).to(dev)
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
            us = ms / max(n, 1) * 1e3
            frac = flop / (us * 1e-6) / PEAK
            if label.endswith("(the family)"):
                family[opt_name] = us / flop
            rel = (us / flop) / family[opt_name] if opt_name in family else float("nan")
            print("| %s | %s | `%s` | %.1f | %.3f | %.1f | %.2f |" % (label, opt_name, name, us, frac, step_us, rel), flush=True)
            del eng


if __name__ == "__main__":
    main()
