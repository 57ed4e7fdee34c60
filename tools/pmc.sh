#!/bin/bash
# tools/pmc.sh -- rocprofv3 PMC passes (counters only; no trace domains) over tools/kbench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  OUT=$R/gpurun_out/pmc_${TAG:-x}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/tools/kbench.py > $OUT/log.txt 2>&1
  echo "== pass $i [$set] rc=$?"
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f
done
