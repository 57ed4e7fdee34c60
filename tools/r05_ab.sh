#!/bin/bash
# tools/r05_ab.sh -- round-5 A/B of prebuilt library variants on ONE box: bench (3 frames in flight, then one) per variant, then the
# SQ instruction counters of every launch in the TIMED geometry (tools/dcstep3.py) per variant.
#   VARIANTS="R4 E1" PMCV="orig R4" bash tools/r05_ab.sh        results: gpurun_out/r05_ab_<TAG>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_ab_${TAG:-x}.txt
: > $OUT
cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
use() { [ $1 = orig ] && cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so || cp $R/link_amd/lib/variants/lib_$1.so $R/link_amd/lib/liblink_amd.so; }
for rep in 1 2; do
for v in orig ${VARIANTS:-}; do
  use $v
  echo "== variant $v (pass $rep)" | tee -a $OUT
  [ $rep = 1 ] && timeout 120 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT
  for st in 3 1; do
    timeout 200 python $R/bench.py --steps 300 --warmup 20 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   streams $st: %.2f us/frame  kernels %s  whole-step frac %.3f single-frame median %.2f' % (d['us_per_frame'], r.get('kernel_us'), r['whole_step']['frac'], r['single_frame_step']['median_us']))
" | tee -a $OUT
  done
done
done
for v in ${PMCV:-}; do
  use $v
  for NS in 3 1; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
    i=$((i+1))
    D=/tmp/pmc_$v_$i; rm -rf $D; mkdir -p $D
    DC_STREAMS=$NS DC_STEPS=40 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/dcstep3.py > $D/log.txt 2>&1
    echo "== pmc variant $v streams $NS pass $i [$set] rc=$?" | tee -a $OUT
    f=$(find $D -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f | tee -a $OUT
    rm -rf $D
  done
  done
done
cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
