#!/usr/bin/env python
"""What a user-supplied device function costs (run on a GPU box): Pendulum, CEM N=500 H=30 5 iterations (config 2's
size) with (a) the built-in fused kernel, (b) the built-in per-iteration kernels, (c) user reward + user dynamics in the
hiprtc-compiled fused rollout kernel, (d) the same through the step-wise evaluator.  us per control step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rate(eng, start, steps=200):
    import torch
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(start).to(dev)
    nx = torch.empty_like(st)
    rec = torch.zeros((start.shape[0], 5), device=dev)
    for _ in range(10):
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
    from blackbox_mpc_amd.utils import synthetic as SY
    from test_gpu_user_functions import INTENDED_PENDULUM_REWARD, USER_PENDULUM_MODEL
    kw = dict(dim_s=3, num_agents=1, planning_horizon=30, population_size=500, max_iterations=5, num_elite=50)
    start = SY.pendulum_start_states(1)
    out = {}
    out["built-in, persistent kernel"] = rate(Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], **kw), start)
    os.environ["BBMPC_FUSED"] = "0"
    out["built-in, per-iteration kernels"] = rate(Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], **kw), start)
    del os.environ["BBMPC_FUSED"]
    for name, env in (("user reward + user dynamics, fused hiprtc rollout", None), ("user reward + user dynamics, step-wise", "1")):
        if env:
            os.environ["BBMPC_USER_STEPWISE"] = env
        e = Engine(L.OPT_CEM, L.DYN_USER, L.REW_USER, [-2.0], [2.0], **kw)
        os.environ.pop("BBMPC_USER_STEPWISE", None)
        e.set_reward_source(INTENDED_PENDULUM_REWARD)
        e.set_dynamics_source(USER_PENDULUM_MODEL)
        out[name] = rate(e, start, 100)
    # the common case: learned MLP dynamics + a custom reward (HalfCheetah MLP, CEM N=1000 H=30 5 iterations)
    from test_gpu_user_functions import USER_CHEETAH_REWARD
    S, U = 20, 6
    kw = dict(dim_s=S, num_agents=1, planning_horizon=30, population_size=1000, max_iterations=5, num_elite=50)
    cstart = SY.cheetah_start_states(1)

    def cheetah(rew, env=None):
        if env:
            os.environ["BBMPC_USER_STEPWISE"] = env
        e = Engine(L.OPT_CEM, L.DYN_MLP, rew, [-1.0] * U, [1.0] * U, **kw)
        os.environ.pop("BBMPC_USER_STEPWISE", None)
        e.set_mlp(*SY.make_mlp_params(), [1, 1, 0], SY.cheetah_stats(S, U))
        if rew == L.REW_USER:
            e.set_reward_source(USER_CHEETAH_REWARD)
        return e

    def rate27(eng, steps):
        import torch
        dev = torch.device("cuda", 0)
        st = torch.from_numpy(cstart).to(dev)
        nx = torch.empty_like(st)
        rec = torch.zeros((1, 27), device=dev)
        for _ in range(5):
            eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
            st, nx = nx, st
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.optimize_dev(st.data_ptr(), rec.data_ptr(), d_next_state=nx.data_ptr())
            st, nx = nx, st
        eng.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6
    out["MLP + built-in cheetah reward (4-particle MFMA kernel)"] = rate27(cheetah(L.REW_CHEETAH), 100)
    out["MLP + user reward: MFMA rollout records the trajectory, one scoring launch"] = rate27(cheetah(L.REW_USER), 100)
    out["MLP + user reward, step-wise"] = rate27(cheetah(L.REW_USER, "1"), 20)
    for k, v in out.items():
        print("| %s | %.1f |" % (k, v))


if __name__ == "__main__":
    main()
