#!/usr/bin/env python
"""f-4 measured: does sharding ONE agent's population over G GPUs beat one GPU?  (MLP path)

usage: popshard_table.py [PI2|CEM|SPSA|PSO|CMAES|RS ...]   (default: PI2)

Run on a GPU box.  For each total population N and shard count G it times one rank's share of a control step --
N/G particles, 5 PI2 iterations, each with the partial refit, the exchange and the merge -- on one MI355X:
  local    : the shard's own work, exchange replaced by a device copy           (BBMPC_POPSHARD_FORCE, no communicator)
  +rccl(1) : the same with ncclAllGather on the launch stream in a ONE-rank communicator: what the collective's launch
             and completion cost the stream before any link latency is added
A G-rank run costs at least  t_local(N/G) + iters * (t_rccl1 - t_local at G=1 ... measured per iteration) + link latency;
the table prints the implied break-even: the largest per-iteration collective latency at which G ranks still beat one.
Multi-GPU latency itself cannot be measured on a one-GPU box; xGMI all-gathers of ~1 KB are latency bound (tens of us)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


OPT = "PI2"


def _opt_kwargs(L):
    code = {"PI2": L.OPT_PI2, "CEM": L.OPT_CEM, "SPSA": L.OPT_SPSA, "PSO": L.OPT_PSO, "CMAES": L.OPT_CMAES,
            "RS": L.OPT_RANDOM_SEARCH}[OPT]
    kw = dict(max_iterations=5, seed=0)
    if OPT == "PI2":
        kw["lamda"] = 1.0
    if OPT in ("CEM", "CMAES"):
        kw["num_elite"] = 50
    if OPT == "CMAES":
        kw["quirks"] = L.CMAES_PER_AGENT
    if OPT == "RS":
        kw["max_iterations"] = 1
    return code, kw


def engine(L, N, Ntot, rccl):
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    S, U = 20, 6
    os.environ["BBMPC_POPSHARD_FORCE"] = "1"
    code, kw = _opt_kwargs(L)
    eng = Engine(code, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=30,
                 population_size=N, population_global=N, **kw)
    del os.environ["BBMPC_POPSHARD_FORCE"]
    eng.set_mlp(*SY.make_mlp_params(), [1, 1, 0], SY.cheetah_stats(S, U))
    if rccl:
        eng.comm_init(Engine.comm_unique_id(), 1, 0)
    return eng


def plain(L, N):
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    S, U = 20, 6
    code, kw = _opt_kwargs(L)
    eng = Engine(code, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=30,
                 population_size=N, **kw)
    eng.set_mlp(*SY.make_mlp_params(), [1, 1, 0], SY.cheetah_stats(S, U))
    return eng


def rate(eng, steps=60):
    import torch
    from blackbox_mpc_amd.utils import synthetic as SY
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(SY.cheetah_start_states(1)).to(dev)
    nx = torch.empty_like(st)
    rec = torch.zeros((1, 27), device=dev)
    for _ in range(20):
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
    global OPT
    from blackbox_mpc_amd import _build
    _build.build()
    from blackbox_mpc_amd import _lib as L
    for OPT in (sys.argv[1:] or ["PI2"]):
        table(L)


def table(L):
    iters = 1 if OPT == "RS" else 5
    print("\n### %s (HalfCheetah MLP 26-200-200-20, H = 30, %d iteration(s), one agent)\n" % (OPT, iters))
    print("| N total | G | particles/rank | one GPU, unsharded (us/step) | shard local (us) | shard + 1-rank ncclAllGather (us) | "
          "collective floor per iteration (us) | break-even link latency per iteration (us) |")
    print("|---|---|---|---|---|---|---|---|")
    for Ntot in (1000, 4000, 8000):
        base = rate(plain(L, Ntot))
        for G in (1, 2, 4, 8):
            n = Ntot // G
            loc = rate(engine(L, n, Ntot, False))
            e = engine(L, n, Ntot, True)
            rc = rate(e)
            e.synchronize()
            e.comm_destroy()
            floor = (rc - loc) / iters
            be = (base - rc) / iters
            print("| %d | %d | %d | %.1f | %.1f | %.1f | %.1f | %s |" % (Ntot, G, n, base, loc, rc, floor,
                                                                      ("%.1f" % be) if G > 1 else "-"))


if __name__ == "__main__":
    main()
