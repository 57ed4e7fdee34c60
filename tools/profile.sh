#!/bin/bash
# tools/profile.sh -- rocprofv3 kernel trace + stats of the bench command; summary copied to gpurun_out/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-run}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-1500
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "stats: $f"; head -25 $f
