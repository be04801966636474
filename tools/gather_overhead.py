"""What the per-control-step all-gather costs next to a ~50 us control step (one-rank RCCL group on one GPU).

Prints, for several ways of handing the record to the collective, the host time of the loop (before the final
synchronize) and the total time per step.  python tools/gather_overhead.py [steps]
"""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from blackbox_mpc_amd import _lib as L
from blackbox_mpc_amd.engine import Engine
from oracle import oracle_np as O


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    eng = Engine(L.OPT_CEM, L.DYN_PENDULUM, L.REW_PENDULUM, [-2.0], [2.0], dim_s=3, num_agents=1, planning_horizon=30,
                 population_size=500, max_iterations=5, num_elite=50, seed=0, device=0)
    eng.reset()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng.set_torch_stream(stream)
    state = torch.from_numpy(O.pendulum_start_states(1)).to(dev)
    nxt = torch.empty_like(state)
    records = [torch.zeros((1, 5), device=dev) for _ in range(2)]
    gathered = [torch.zeros((1, 5), device=dev) for _ in range(2)]
    comm = torch.cuda.Stream(device=dev)
    ev_r = [torch.cuda.Event() for _ in range(2)]
    ev_d = [torch.cuda.Event() for _ in range(2)]

    from blackbox_mpc_amd.parallel import attach_record_comm
    attach_record_comm(eng, device=dev)

    def run(mode, every=1):
        nonlocal state, nxt
        works = [None, None]
        pend = [False, False]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            b = i & 1
            if works[b] is not None:
                if mode == "async_query":
                    if not works[b].is_completed():
                        works[b].wait()
                else:
                    works[b].wait()
                works[b] = None
            if mode in ("native", "native_fused"):
                eng.gather_wait(b)
            if mode == "native_fused":
                eng.optimize_gather_dev(state.data_ptr(), records[b].data_ptr(), gathered[b].data_ptr(), b,
                                        d_next_state=nxt.data_ptr())
                state, nxt = nxt, state
                continue
            if pend[b]:
                if mode == "events_query":
                    if not ev_d[b].query():
                        stream.wait_event(ev_d[b])
                else:
                    stream.wait_event(ev_d[b])
                pend[b] = False
            eng.optimize_dev(state.data_ptr(), records[b].data_ptr(), d_next_state=nxt.data_ptr())
            if i % every == 0:
                if mode in ("async", "async_query"):
                    works[b] = dist.all_gather_into_tensor(gathered[b], records[b], async_op=True)
                elif mode in ("events", "events_query"):
                    ev_r[b].record(stream)
                    comm.wait_event(ev_r[b])
                    with torch.cuda.stream(comm):
                        dist.all_gather_into_tensor(gathered[b], records[b])
                        ev_d[b].record(comm)
                    pend[b] = True
                elif mode == "native":
                    eng.gather_records_dev(records[b].data_ptr(), gathered[b].data_ptr(), 5, b)
                elif mode == "comm_only":          # unordered copy on the side stream: cost of concurrency alone
                    with torch.cuda.stream(comm):
                        gathered[b].copy_(records[b])
                elif mode == "comm_only_nccl":
                    with torch.cuda.stream(comm):
                        dist.all_gather_into_tensor(gathered[b], records[b])
                elif mode == "inline":
                    dist.all_gather_into_tensor(gathered[b], records[b])
                elif mode == "record_only":
                    ev_r[b].record(stream)
            state, nxt = nxt, state
        t1 = time.perf_counter()
        eng.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mode:14s} every {every:2d}: host {1e6 * (t1 - t0) / steps:6.1f} us/step   total {1e6 * (t2 - t0) / steps:6.1f} us/step",
              flush=True)

    for _ in range(2):
        run("none")
    for mode in ("native", "native_fused", "comm_only", "comm_only_nccl", "record_only", "async", "async_query", "events", "events_query", "inline"):
        run(mode)
    run("native")
    run("native_fused")
    assert torch.equal(gathered[0], records[0]) and torch.equal(gathered[1], records[1])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
