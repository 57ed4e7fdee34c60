#!/bin/bash
# tools/pmc_dc.sh -- rocprofv3 PMC passes (counters only; no trace domains besides --kernel-trace) over
# tools/dcstep.py: TAG=name bash tools/pmc_dc.sh "SET1" "SET2" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  OUT=$R/gpurun_out/pmc_${TAG:-dc}_$i
  rm -rf $OUT; mkdir -p $OUT
  DC_STEPS=${DC_STEPS:-60} timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/${SCRIPT:-tools/dcstep.py} > $OUT/log.txt 2>&1
  echo "== pass $i [$set] rc=$?"
  f=$(find $OUT -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f
  rm -f $f $(find $OUT -name "*.csv" | head -20)
done
