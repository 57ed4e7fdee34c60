#!/bin/bash
# tools/ab_bench.sh -- bench.py (3 frames in flight, then one) + smoke for prebuilt library variants: VARIANTS="E F" bash tools/ab_bench.sh
R=$GRAFT_REPO_ROOT
cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
for v in orig ${VARIANTS:-}; do
  [ $v = orig ] && cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so || cp $R/link_amd/lib/variants/lib_$v.so $R/link_amd/lib/liblink_amd.so
  echo "== variant $v"
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  for st in 3 1; do
    timeout 200 python $R/bench.py --steps 200 --warmup 20 --streams $st --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('   streams $st: %.2f us/frame  value %.3e  kernels %s  whole-step frac %.3f' % (d['us_per_frame'], d['value'], r.get('kernel_us'), r['whole_step']['frac']))
"
  done
done
cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
