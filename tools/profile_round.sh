#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE /
# SQ busy counters) for the bench configurations.  Raw output goes to gpurun_out/prof_$TAG, the summaries that
# get committed are written by tools/summarize_profiles.py into profiles/.
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
for CFG in cfg2 cfg3 cfg4 cfg5cem; do
  STEPS=200; [ $CFG = cfg4 ] && STEPS=30; [ $CFG = cfg5cem ] && STEPS=10
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/$CFG -o trace -- python bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline > $OUT/$CFG.bench.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/$CFG -o fetch -- python bench.py --config $CFG --steps 20 --warmup 2 --no-cpu-baseline > $OUT/$CFG.fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/$CFG -o write -- python bench.py --config $CFG --steps 20 --warmup 2 --no-cpu-baseline > $OUT/$CFG.write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -f csv -d $OUT/$CFG -o sq -- python bench.py --config $CFG --steps 20 --warmup 2 --no-cpu-baseline > $OUT/$CFG.sq.log 2>&1
  grep "^{\"metric\"" $OUT/$CFG.bench.log | tail -1 > $OUT/$CFG.bench.json
done
python tools/summarize_profiles.py $OUT $TAG
ls -la profiles/ gpurun_out/profiles_$TAG 2>/dev/null | head -30
