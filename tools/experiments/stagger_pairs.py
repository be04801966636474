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
    pair = [bench.Workload(name, r, 2, 0, dev, False, "nccl", "none", agents=2) for r in range(2)]
    loop(pair, 3, 0.0)
    for d in delays:
        print("two handles x 2 agents, second started %.2f ms late: %.4f ms per control step" % (d, loop(pair, steps, d * 1e-3)))
    if only == "pair":
        return
    quad = [bench.Workload(name, r, 4, 0, dev, False, "nccl", "none", agents=1) for r in range(4)]
    loop(quad, 3, 0.0)
    for d in delays:
        print("four handles x 1 agent, each started %.2f ms after the previous: %.4f ms per control step" % (d, loop(quad, steps, d * 1e-3)))

if __name__ == "__main__":
    main()
