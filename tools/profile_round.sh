#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE /
# SQ busy counters) for the bench configurations.  Raw output goes to gpurun_out/prof_$TAG, the summaries that
# get committed are written by tools/summarize_profiles.py into gpurun_out/profiles_$TAG (copy them to profiles/).
TAG=${1:-r2}
CFGS=${2:-"cfg2 cfg3 cfg3full cfg4 cfg4pi2 cfg5cem cfg5pso cfg5cma cfg2cma cfg_tut2"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="--no-cpu-baseline --no-secondary"
for CFG in $CFGS; do
  STEPS=200; W=5
  case $CFG in cfg4|cfg4pi2) STEPS=40;; cfg5cem|cfg5pso) STEPS=10;; cfg5cma) STEPS=6;; cfg2cma|cfg_tut2) STEPS=40;; esac
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$CFG -o trace -- python bench.py --config $CFG --steps $STEPS --warmup $W $B > $OUT/$CFG.bench.log 2>&1
  S2=20; [ $CFG = cfg5cma ] && S2=4; [ $CFG = cfg5cem ] && S2=6; [ $CFG = cfg5pso ] && S2=6
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/$CFG -o fetch -- python bench.py --config $CFG --steps $S2 --warmup 2 $B > $OUT/$CFG.fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/$CFG -o write -- python bench.py --config $CFG --steps $S2 --warmup 2 $B > $OUT/$CFG.write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $OUT/$CFG -o sq -- python bench.py --config $CFG --steps $S2 --warmup 2 $B > $OUT/$CFG.sq.log 2>&1
  grep "^{\"metric\"" $OUT/$CFG.bench.log | tail -1 > $OUT/$CFG.bench.json
  # the raw per-dispatch traces are large: keep the stats + counter tables only
  rm -f $OUT/$CFG/*kernel_trace.csv
done
python tools/summarize_profiles.py $OUT $TAG
ls -la gpurun_out/profiles_$TAG 2>/dev/null | head -30
