#!/bin/bash
# tools/profile.sh -- rocprofv3 kernel trace of the bench command (hard timeout: rocprofv3 can hang at
# process exit after writing its output); summary via tools/rocpd_stats.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-run}
rm -rf $OUT; mkdir -p $OUT
timeout ${TMO:-150} rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-50} --warmup ${WARM:-5} --no-cpu-baseline ${ARGS:-} > $OUT/bench.log 2>&1
echo "rocprof rc=$?"
grep '^{' $OUT/bench.log | cut -c1-1200
db=$(find $OUT -name "*.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $OUT/kernel_stats.csv | head -12
