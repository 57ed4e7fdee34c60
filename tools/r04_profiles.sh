#!/bin/bash
# tools/r04_profiles.sh -- everything profiles/r04_* is made from, one gpurun call:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (3 frames in flight) and of --streams 1
#   2. the default bench line, the half-row line and the labelled cfg3 / cfg5 lines (these carry per-stage R_core rooflines)
#   2b. R_core on the LiDAR stage frames: tools/lidar_core.py table + kernel traces of two stages
#   3. PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit+miss / SQ; counters only, separate passes) over tools/dcstep.py
#      (cold steps, one stream) -> pmc_counters.txt -> traffic.json (tools/traffic_json.py)
# TAG=<name> bash tools/r04_profiles.sh ; results under gpurun_out/r04_<TAG>/  (copy into profiles/ as r04_<TAG>_*)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_${TAG:-x}
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
tail -c 300 $OUT/bench_default.json
timeout 200 python $R/bench.py --io f16 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/bench_f16.json
timeout 300 python $R/bench.py --workload cfg3 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg3.json
timeout 300 python $R/bench.py --workload cfg5 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5.json
timeout 300 python $R/bench.py --workload cfg5 --io f16 --steps 30 --warmup 3 2>/dev/null | grep '^{' > $OUT/bench_cfg5_f16.json
# R_core on the LiDAR stage frames (cfg3 / cfg5 stages, tools/lidar_core.py): event-timed table over the 8 stages, both forms
# of the general layout, then kernel traces of the tile form on a cfg3 and a cfg5 stage
timeout 400 python $R/tools/lidar_core.py 2>/dev/null | grep '^{' > $OUT/lidar_stages.jsonl
for s in 0 4; do
  D=$OUT/trace_lidar$s; mkdir -p $D
  FORM=tiles STAGE=$s ITERS=100 timeout 150 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/tools/lidar_core.py > $D/log.txt 2>&1
  db=$(find $D -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_lidar_stage$s.csv | grep -E "k_elk|k_cell_c|k_cell_s|k_place|k_sort" | head -6
  rm -rf $D
done
# the lean form (three launches, index rebuilt) on a small, a mid and a detection stage
for s in 3 1 6; do
  D=$OUT/trace_lean$s; mkdir -p $D
  FORM=lean STAGE=$s ITERS=200 timeout 150 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $R/tools/lidar_core.py > $D/log.txt 2>&1
  db=$(find $D -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_stats.py $db $OUT/kernel_stats_lean_stage$s.csv | grep -E "k_lean" | head -4
  rm -rf $D
done
# the voxel-scan numbering of the general layout's index against the cell scan (A/B on this box), and the three-frame step
# kernel against three plans on three streams
for o in cell first; do ORDER=$o FORM=tiles timeout 300 python $R/tools/lidar_core.py 2>/dev/null | grep '^{' > $OUT/lidar_stages_order_$o.jsonl; done
FRAMES=600 timeout 200 python $R/tools/step3.py 2>&1 | grep -E "^frame|us/frame" > $OUT/step_kernel_vs_streams.txt
cat $OUT/step_kernel_vs_streams.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  D=$OUT/pmc_$i; mkdir -p $D
  DC_STEPS=60 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/dcstep.py > $D/log.txt 2>&1
  echo "== pass $i [$set] rc=$?" >> $OUT/pmc_counters.txt
  f=$(find $D -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f >> $OUT/pmc_counters.txt
  rm -rf $D
done
cat $OUT/pmc_counters.txt
python $R/tools/traffic_json.py $OUT/pmc_counters.txt $OUT/traffic.json "${COMMIT:-unknown}" "profiles/r04_${TAG:-x}_pmc_counters.txt"
cat $OUT/traffic.json
