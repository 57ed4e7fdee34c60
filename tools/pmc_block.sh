#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_nbr; rm -rf $OUT; mkdir -p $OUT
i=0
# counters of the block driver loop (tools/block_timeline.py), neighbour-map and insert kernels: bash tools/pmc_block.sh
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1)); D=$OUT/p$i; mkdir -p $D
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/${SCRIPT:-tools/block_timeline.py} > $D/log.txt 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  echo "== pass $i [$set]" >> $OUT/summary.txt
  [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -i "${KERNELS:-neighbor_map\|probe_bbox}" >> $OUT/summary.txt
  rm -rf $D
done
cat $OUT/summary.txt
