#!/bin/bash
# tools/ab_step3.sh -- tools/step3.py for prebuilt library variants on one box: VARIANTS="k2p1 k2p3" bash tools/ab_step3.sh
R=$GRAFT_REPO_ROOT
cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
for rep in 1 2; do
for v in orig ${VARIANTS:-}; do
  [ $v = orig ] && cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so || cp $R/link_amd/lib/variants/lib_$v.so $R/link_amd/lib/liblink_amd.so
  echo "== variant $v"
  FRAMES=${FRAMES:-600} timeout 200 python $R/tools/step3.py 2>&1 | grep "us/frame\|equal = False" | sed -n '2,3p;5p'
done; done
cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
