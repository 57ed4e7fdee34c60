#!/bin/bash
# tools/r05_profiles.sh -- everything profiles/r05_* is made from, one gpurun call (the round-4 script + the SQ / byte counters in the
# launch geometry bench.py TIMES, tools/dcstep3.py, next to the one-frame geometry, tools/dcstep.py):
#   TAG=<name> COMMIT=<sha> bash tools/r05_profiles.sh ; results under gpurun_out/r05_<TAG>/  (copy into profiles/ as r05_<TAG>_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
for ST in 3 1; do
  D=$OUT/trace_s$ST; mkdir -p $D
  timeout 200 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/bench.py --steps 100 --warmup 10 --streams $ST --no-cpu-baseline > $D/bench.log 2>&1
  grep '^{' $D/bench.log > $OUT/bench_under_rocprof_streams$ST.json
  db=$(find $D -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_streams$ST.csv | head -6
  rm -rf $D
done
python $R/bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_driver_args.json
timeout 200 python $R/bench.py --io f16 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_f16.json
timeout 300 python $R/bench.py --workload cfg3 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg3.json
timeout 300 python $R/bench.py --workload cfg5 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5.json
timeout 300 python $R/bench.py --workload cfg5 --io f16 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5_f16.json
timeout 400 python $R/tools/lidar_core.py 2>/dev/null | grep '^{' > $OUT/lidar_stages.jsonl
timeout 400 python $R/tools/lidar_core_parity.py 2>/dev/null | grep '^{' > $OUT/lidar_core_parity.jsonl
for NS in 3 1; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    D=$OUT/pmc_${NS}_$i; mkdir -p $D
    DC_STREAMS=$NS DC_STEPS=40 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/dcstep3.py > $D/log.txt 2>&1
    echo "== pass $i [$set] rc=$?" >> $OUT/pmc_counters_streams$NS.txt
    f=$(find $D -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f >> $OUT/pmc_counters_streams$NS.txt
    rm -rf $D
  done
done
python $R/tools/traffic_json.py $OUT/pmc_counters_streams1.txt $OUT/traffic.json "${COMMIT:-unknown}" "profiles/r05_${TAG:-x}_pmc_counters_streams1.txt"
cat $OUT/traffic.json
