#!/bin/bash
# tools/k1form.sh -- bench.py over the forms of the fused pre_mix kernel (0 cell-range, 1 tile, 2 matrix-core sums) x frames in flight
R=${GRAFT_REPO_ROOT:-.}
for form in ${FORMS:-0 2}; do for st in ${STREAMS:-3 1}; do for wgs in ${WGS:-0}; do
  LINK_BENCH_K1_FORM=$form LINK_BENCH_K1_WGS=$([ $wgs = 0 ] && echo "" || echo $wgs) timeout 300 python $R/bench.py --steps 200 --warmup 20 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   k1_form $form streams $st k1_wgs $wgs: %.2f us/frame  kernels %s  single-frame median %.2f  frac %.3f  check %s' % (d['us_per_frame'], r.get('kernel_us'), r['single_frame_step']['median_us'], r['whole_step']['frac'], d.get('timed_configuration_check')))
"
done; done; done
