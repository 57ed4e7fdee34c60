#!/bin/bash
# tools/r02_profiles.sh -- everything profiles/r02_* is made from, one gpurun call:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (3 frames in flight) and of --streams 1
#   2. PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit+miss / SQ) over tools/dcstep.py (cold steps, one stream)
# TAG=<name> bash tools/r02_profiles.sh ; results under gpurun_out/r02_<TAG>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
for ST in 3 1; do
  D=$OUT/trace_s$ST; mkdir -p $D
  timeout 200 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/bench.py --steps 100 --warmup 10 --streams $ST --no-cpu-baseline > $D/bench.log 2>&1
  grep '^{' $D/bench.log > $OUT/bench_under_rocprof_streams$ST.json
  db=$(find $D -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_streams$ST.csv | head -8
  rm -rf $D
done
python $R/bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  D=$OUT/pmc_$i; mkdir -p $D
  DC_STEPS=60 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/dcstep.py > $D/log.txt 2>&1
  echo "== pass $i [$set] rc=$?" >> $OUT/pmc_counters.txt
  f=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f >> $OUT/pmc_counters.txt
  rm -rf $D
done
cat $OUT/pmc_counters.txt
