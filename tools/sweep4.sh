#!/bin/bash
# tools/sweep4.sh -- bench.py over frames in flight x z-segments of the gather kernel x form / waves of the fused pre_mix kernel
R=${GRAFT_REPO_ROOT:-.}
for st in ${STREAMS:-3 4 6}; do for zs in ${ZS:-1 2 3}; do for form in ${FORMS:-0 2}; do for wgs in ${WGS:-256 512}; do
  LINK_BENCH_K1_FORM=$form LINK_BENCH_K1_WGS=$wgs LINK_BENCH_K2_ZSPLIT=$zs timeout 200 python $R/bench.py --steps ${STEPS:-150} --warmup 10 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   streams $st zsplit $zs k1_form $form k1_wgs $wgs: %.2f us/frame (events %.2f)  frac %.3f' % (d['us_per_frame'], 1e3 * d['ms_per_step_event_median'] / $st, r['whole_step']['frac']))
"
done; done; done; done
