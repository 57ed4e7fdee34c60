#!/bin/bash
# tools/k1ab.sh -- bench.py with the cell-range form (k1_form 0) and the tile form (k1_form 1) of the fused pre_mix kernel, 3 and 1 frames in flight
R=${GRAFT_REPO_ROOT:-.}
for form in ${FORMS:-0 1}; do
  for st in ${STREAMS:-3 1}; do
    LINK_BENCH_K1_FORM=$form timeout 300 python $R/bench.py --steps 200 --warmup 20 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   k1_form $form streams $st: %.2f us/frame  value %.3e  kernels %s  single-frame median %.2f  whole-step frac %.3f  check %s' % (d['us_per_frame'], d['value'], r.get('kernel_us'), r['single_frame_step']['median_us'], r['whole_step']['frac'], d.get('timed_configuration_check')))
"
  done
done
