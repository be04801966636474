#!/usr/bin/env python
"""Host enqueue time vs GPU drain time of one control step (device-resident entry point, empty queue at the start).

usage: enqueue_probe.py cfg2cma [cfg5cma ...]
Prints, per configuration: the time bbmpc_optimize_dev takes to return (the host's launch work), the time until the
stream is idle after that, and the median of MPCPolicy.act for comparison.  If enqueue + drain ~ act and drain is well
under the device-resident step time, the host is the bottleneck (launch-bound path)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    names = sys.argv[1:] or ["cfg2cma"]
    for name in names:
        W = bench.Workload(name, 0, 1, 0, torch.device("cuda:0"), False, "nccl", "async")
        for _ in range(5):
            W.dev_step()
        W.fence()
        enq, drain = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            W.dev_step()
            t1 = time.perf_counter()
            W.eng.synchronize()
            t2 = time.perf_counter()
            enq.append(t1 - t0)
            drain.append(t2 - t1)
        W.fence()
        t0 = time.perf_counter()
        for _ in range(30):
            W.dev_step()
        W.fence()
        back = (time.perf_counter() - t0) / 30
        acts = []
        for t in range(30):
            t0 = time.perf_counter()
            W.act_step(t)
            acts.append(time.perf_counter() - t0)
        print("%-8s enqueue %.1f us   drain after enqueue %.1f us   sum %.1f us | back-to-back %.1f us/step | act median %.1f us"
              % (name, np.median(enq) * 1e6, np.median(drain) * 1e6, (np.median(enq) + np.median(drain)) * 1e6, back * 1e6,
                 np.median(acts) * 1e6))
        W.close()


if __name__ == "__main__":
    main()
