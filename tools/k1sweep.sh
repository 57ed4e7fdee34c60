#!/bin/bash
# tools/k1sweep.sh -- frames in flight x launch geometry of the fused pre_mix kernel through bench.py (per-plan tuning via LINK_BENCH_*)
R=${GRAFT_REPO_ROOT:-.}
for k2 in ${K2FORMS:-0}; do for form in ${FORMS:-0}; do for st in ${STREAMS:-3 4}; do for wgs in ${WGS:-256 384 512 768}; do for zs in ${ZS:-2}; do
  LINK_BENCH_K2_FORM=$k2 LINK_BENCH_K1_FORM=$form LINK_BENCH_K1_WGS=$wgs LINK_BENCH_K2_ZSPLIT=$zs timeout 300 python $R/bench.py --steps 300 --warmup 30 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   k2_form $k2 k1_form $form streams $st k1_wgs $wgs zsplit $zs: %.2f us/frame  frac %.3f' % (d['us_per_frame'], r['whole_step']['frac']))
"
done; done; done; done; done
