#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_blk
rm -rf $OUT; mkdir -p $OUT
python $GRAFT_REPO_ROOT/tools/block_timeline.py > $OUT/plain.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $GRAFT_REPO_ROOT/tools/block_timeline.py > $OUT/run.log 2>&1
grep -v amdgpu.ids $OUT/plain.log | tail -2
db=$(find $OUT -name "*.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/block_timeline.py $db > $OUT/timeline.txt 2>&1 && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $OUT/kernel_stats.csv > /dev/null
rm -f $db
cat $OUT/timeline.txt
