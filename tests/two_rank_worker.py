"""Worker of tests/test_gpu_two_ranks.py: one of TWO ranks (one GPU each) started by torch.distributed.run.  Exercises the
engine's RCCL paths with nranks = 2 and writes what it saw to $BBMPC_TWO_RANK_OUT.rank<r>.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    assert world == 2 or os.environ.get("BBMPC_TWO_RANK_ALLOW_ONE")       # (one rank: the script's own smoke test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from blackbox_mpc_amd import _lib as L
    from blackbox_mpc_amd import parallel as P
    from blackbox_mpc_amd.engine import Engine
    from blackbox_mpc_amd.utils import synthetic as SY
    out = {"rank": rank}

    # ---- (i) agent shards + bbmpc_optimize_gather: bit-equal to the unsharded engine, per agent, 3 closed-loop steps
    A_glob, N, H, iters = 4, 300, 20, 3
    off, cnt = P.agent_shard(A_glob, world, rank)

    def pend(opt, A, **kw):
        return Engine(opt, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=A, planning_horizon=H,
                      population_size=N, max_iterations=iters, num_elite=30, seed=5, device=local, **kw)

    worst = 0.0
    for opt in (L.OPT_CEM, L.OPT_PI2):
        full = pend(opt, A_glob)
        mine = pend(opt, cnt, agent_offset=off, num_agents_global=A_glob)
        P.attach_record_comm(mine, device=dev)
        out["rccl_ranks"] = mine.comm_info()[0]
        rec_w = 1 + 3 + 1
        gathered = [torch.full((A_glob, rec_w), -7.0, device=dev) for _ in range(2)]
        state = SY.pendulum_start_states(A_glob)
        for t in range(3):
            a_f, n_f, r_f = full.optimize(state)
            a_m, n_m, r_m = mine.optimize_gather(state[off:off + cnt], gathered[t & 1].data_ptr(), t & 1)
            mine.gather_wait(t & 1, host_block=True)
            g = gathered[t & 1].cpu().numpy()
            want = np.concatenate([a_f, n_f, np.asarray(r_f).reshape(-1, 1)], axis=1)
            worst = max(worst, float(np.abs(g - want).max()))                       # every rank holds every agent's record
            assert np.array_equal(a_m, a_f[off:off + cnt]) and np.array_equal(n_m, n_f[off:off + cnt])
            state = n_f
        mine.comm_destroy()
    out["agent_shard_gather_max_abs_diff"] = worst

    # ---- (ii) population shards of ONE agent (PI2, CEM) at the north-star shape against the unsharded engine
    S, U, Hm, Nm = 20, 6, 30, 1000
    ws, bs = SY.make_mlp_params()
    stats = SY.cheetah_stats(S, U)
    st = SY.cheetah_start_states(1, S)
    pop = {}
    for name, opt in (("PI2", L.OPT_PI2), ("CEM", L.OPT_CEM)):
        def mk(n, **kw):
            e = Engine(opt, L.DYN_MLP, L.REW_CHEETAH, [-1.0] * U, [1.0] * U, dim_s=S, num_agents=1, planning_horizon=Hm,
                       population_size=n, max_iterations=5, num_elite=50, seed=9, device=local, **kw)
            e.set_mlp(ws, bs, [L.ACT_TANH, L.ACT_TANH, L.ACT_NONE], stats)
            return e
        one = mk(Nm)
        p_off, p_cnt = P.population_shard(Nm, world, rank)
        shard = mk(p_cnt, population_offset=p_off, population_global=Nm)
        P.attach_record_comm(shard, device=dev)
        d = 0.0
        s_one, s_sh = st.copy(), st.copy()
        for t in range(2):
            a1, n1, _ = one.optimize(s_one)
            a2, n2, _ = shard.optimize(s_sh)
            d = max(d, float(np.abs(a1 - a2).max()), float(np.abs(n1 - n2).max()))
            s_one, s_sh = n1, n2
        pop[name] = d
        shard.comm_destroy()
    out["popshard_max_abs_diff"] = pop
    with open("%s.rank%d.json" % (os.environ["BBMPC_TWO_RANK_OUT"], rank), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
