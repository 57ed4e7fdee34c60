#!/bin/bash
# tools/r06_profiles.sh -- everything profiles/r06_* is made from, one gpurun call:
#   TAG=<name> COMMIT=<sha> bash tools/r06_profiles.sh ; results under gpurun_out/r06_<TAG>/  (copy into profiles/ as r06_<TAG>_*)
# kernel trace + stats of the bench command (3 and 1 frames in flight), PMC passes (counters only, one set per pass) over the step in
# the TIMED geometry, profiles/timed_geometry.json, the default / driver-args / cfg3 / cfg5 bench lines, the batch entry point's
# trace (three persistent kernels) and timeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
for ST in 3 1; do
  D=$OUT/trace_s$ST; mkdir -p $D
  timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/bench.py --steps 100 --warmup 10 --streams $ST --no-cpu-baseline > $D/bench.log 2>&1
  grep '^{' $D/bench.log > $OUT/bench_under_rocprof_streams$ST.json
  db=$(find $D -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_streams$ST.csv | head -6
  rm -rf $D
done
for NS in 3; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    D=$OUT/pmc_${NS}_$i; mkdir -p $D
    DC_STREAMS=$NS DC_STEPS=40 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/dcstep3.py > $D/log.txt 2>&1
    echo "== pass $i [$set] rc=$?" >> $OUT/pmc_counters_streams$NS.txt
    f=$(find $D -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f >> $OUT/pmc_counters_streams$NS.txt
    rm -rf $D
  done
done
python $R/tools/timed_geometry_json.py $OUT/kernel_stats_streams3.csv $OUT/pmc_counters_streams3.txt $OUT/timed_geometry.json "${COMMIT:-unknown}" "profiles/r06_${TAG:-x}"
cp $OUT/timed_geometry.json $R/profiles/timed_geometry.json      # the line below is built from THIS run's profile
python $R/bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_driver_args.json
if [ -z "$QUICK" ]; then
timeout 300 python $R/bench.py --workload cfg3 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg3.json
timeout 300 python $R/bench.py --workload cfg5 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5.json
timeout 300 python $R/bench.py --workload cfg5 --io f16 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5_f16.json
timeout 200 python $R/bench.py --io f16 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_f16.json
# row N1: kernel trace of the cfg3 line (the convolution kernels' averages next to the line's conv_roofline) and of the training step
D=$OUT/trace_cfg3; mkdir -p $D
timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/bench.py --workload cfg3 --steps 30 --warmup 3 > /dev/null 2>&1
db=$(find $D -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/cfg3_conv_kernel_stats.csv | head -5; rm -rf $D
D=$OUT/trace_train; mkdir -p $D
K=15 REPS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/tools/cfg3train_prof.py > $OUT/cfg3_train.log 2>&1
db=$(find $D -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/cfg3_train_kernel_stats.csv | head -5; rm -rf $D
timeout 400 python $R/tools/lidar_core.py 2>/dev/null | grep '^{' > $OUT/lidar_stages.jsonl
fi
# the batch entry point: kernel trace of its three persistent kernels, and where their workgroups spend the call
D=$OUT/trace_batch; mkdir -p $D
GPU_MAX_HW_QUEUES=8 B=24 SETS=2 STEPS=20 PASSES=1 timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/tools/batch_bench.py > $OUT/batch_bench_under_rocprof.txt 2>&1
db=$(find $D -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_batch.csv | head -8
rm -rf $D
GPU_MAX_HW_QUEUES=8 B=24 SETS=2 timeout 300 python $R/tools/batch_bench.py > $OUT/batch_bench.txt 2>&1
GPU_MAX_HW_QUEUES=8 B=48 SETS=2 STEPS=30 timeout 300 python $R/tools/batch_bench.py > $OUT/batch_bench_b48.txt 2>&1
GPU_MAX_HW_QUEUES=8 B=24 SETS=1 timeout 300 python $R/tools/batch_bench.py > $OUT/batch_bench_1set.txt 2>&1
if [ -f $R/link_amd/lib/variants/lib_BTPROF.so ]; then
  cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
  cp $R/link_amd/lib/variants/lib_BTPROF.so $R/link_amd/lib/liblink_amd.so
  (cd $R && GPU_MAX_HW_QUEUES=8 B=24 timeout 300 python tools/batch_timeline.py > $OUT/batch_timeline.txt 2>&1; cp gpurun_out/batch_timeline.json $OUT/batch_timeline.json)
  cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
fi
