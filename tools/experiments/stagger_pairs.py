"""Experiment (GPU box): config 5 / CMA-ES, the four instances of a GPU as ONE handle against TWO handles of two agents
each (agent_offset 0 / 2: the same random streams, as for ranks of an agent-sharded run), each on its own stream,
started together or half an iteration apart.  Prints ms per control step of all four agents (device-resident loop).
    python tools/experiments/stagger_pairs.py [steps] [delay_ms ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

def loop(ws, steps, delay_s):
    for w in ws:
        w.fence()
    t0 = time.perf_counter()
    for t in range(steps):
        for i, w in enumerate(ws):
            if t == 0 and i > 0 and delay_s > 0:
                time.sleep(delay_s)
            w.dev_step()
    for w in ws:
        w.fence()
    return (time.perf_counter() - t0 - (delay_s if len(ws) > 1 else 0.0)) / steps * 1e3

def masked(on):
    """Round 5: the handles' launch streams on CUs 0 .. 247, the tridiagonalisations on side streams confined to CUs 248 .. 255
    (BBMPC_STREAM_CUS / BBMPC_EIGH_SIDE_CUS, read when a handle is created / when its first decomposition is launched)."""
    split = int(os.environ.get("STAGGER_SPLIT", "248"))
    for k, v in (("BBMPC_STREAM_CUS", "0-%d" % (split - 1)), ("BBMPC_EIGH_SIDE_CUS", "%d-255" % split)):
        if on:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    delays = [float(x) for x in sys.argv[2:]] or [0.0, 0.3, 0.55, 0.8]
    dev = torch.device("cuda:0")
    name = os.environ.get("STAGGER_CFG", "cfg5cma")
    only = os.environ.get("STAGGER_ONLY", "")
    if only in ("", "one"):
        one = bench.Workload(name, 0, 1, 0, dev, False, "nccl", "none")
        loop([one], 3, 0.0)
        print("one handle, 4 agents: %.4f ms per control step" % loop([one], steps, 0.0))
        one.close()
    if only in ("", "half"):
        half = bench.Workload(name, 0, 2, 0, dev, False, "nccl", "none", agents=2)
        loop([half], 3, 0.0)
        print("one handle, 2 agents, alone: %.4f ms per control step" % loop([half], steps, 0.0))
        half.close()
    if only not in ("", "pair"):
        return
    for mask in ((True,) if os.environ.get("STAGGER_MASKED_ONLY") else (False, True)):
        masked(mask)
        tag = ("CU-masked streams (launch 0-%d | tridiagonalisation %d-255)" % (int(os.environ.get("STAGGER_SPLIT", "248")) - 1, int(os.environ.get("STAGGER_SPLIT", "248")))) if mask else "plain streams"
        pair = [bench.Workload(name, r, 2, 0, dev, False, "nccl", "none", agents=2) for r in range(2)]
        loop(pair, 3, 0.0)
        for d in delays:
            print("two handles x 2 agents, %s, second started %.2f ms late: %.4f ms per control step" % (tag, d, loop(pair, steps, d * 1e-3)))
        for w in pair:
            w.close()
        del pair
        if only == "pair":
            continue
        quad = [bench.Workload(name, r, 4, 0, dev, False, "nccl", "none", agents=1) for r in range(4)]
        loop(quad, 3, 0.0)
        for d in delays:
            print("four handles x 1 agent, %s, each started %.2f ms after the previous: %.4f ms per control step" % (tag, d, loop(quad, steps, d * 1e-3)))
        for w in quad:
            w.close()
        del quad
    masked(False)
    # the side stream alone (one handle, all four agents in lock-step): what the two event hand-offs per decomposition cost
    os.environ["BBMPC_EIGH_SIDE_CUS"] = "248-255"
    one = bench.Workload(name, 0, 1, 0, dev, False, "nccl", "none")
    loop([one], 3, 0.0)
    print("one handle, 4 agents, tridiagonalisation on its masked side stream: %.4f ms per control step" % loop([one], steps, 0.0))
    one.close()
    masked(False)

if __name__ == "__main__":
    main()
