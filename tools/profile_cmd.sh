#!/bin/bash
# tools/profile_cmd.sh -- rocprofv3 kernel trace of an arbitrary python script: TAG=name SCRIPT=tools/x.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-cmd}
rm -rf $OUT; mkdir -p $OUT
timeout ${TMO:-150} rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $GRAFT_REPO_ROOT/$SCRIPT ${ARGS:-} > $OUT/run.log 2>&1
echo "rocprof rc=$?"
grep -v amdgpu.ids $OUT/run.log | tail -3
db=$(find $OUT -name "*.db" | head -1)
[ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $OUT/kernel_stats.csv | head -${LINES:-40}
rm -f $db
