#!/bin/bash
# A/B timing of library variants (tools/build_variant.py) on one box: usage  bash tools/pair_ab.sh <config> <rounds> <variant>...
# ("main" = the tree's libbbmpc.so); prints the dominant kernel's average launch time per variant and round.
CFG=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    if [ $v = main ]; then unset BBMPC_LIB; else export BBMPC_LIB=$PWD/tools/variants/libbbmpc_$v.so; fi
    python bench.py --config $CFG --steps 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'round $r', '%s %.1f us  frac %.4f  dev-resident %.4f ms' % (r['kernel'], r['avg_launch_us'], r['frac'], d['device_resident_ms_per_step']))"
  done
done
