#!/bin/bash
# Every bench configuration once (no CPU baseline), one summary line each.  Run on the GPU box: bash tools/bench_all.sh
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%-9s value %10.1f  act median %8.4f ms (p10 %.4f p90 %.4f)  device-resident %8.4f ms  kernel %-22s %8.1f us  %s %.4g %s frac %.4f" % (sys.argv[1], d["value"], d["median_ms"], d["p10_ms"], d["p90_ms"], d["device_resident_ms_per_step"], r["kernel"], r["avg_launch_us"], r["bound"], r["achieved"], r["unit"], r["frac"]))'
for c in cfg1 cfg2 cfg2pi2 cfg2spsa cfg2pso cfg2cma cfg3 cfg3full cfg4 cfg4pi2 cfg5cem cfg5pso cfg5cma cfg5full cfg_tut2; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "$P" $c
done
