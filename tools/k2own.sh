#!/bin/bash
# own-cell gather form: ring depth A/B (prebuilt variants) through tools/k2ab.py and bench.py
R=$GRAFT_REPO_ROOT
cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
for v in orig ${VARIANTS:-RING3}; do
  [ $v = orig ] && cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so || cp $R/link_amd/lib/variants/lib_$v.so $R/link_amd/lib/liblink_amd.so
  echo "== variant $v"
  timeout 200 python $R/tools/k2ab.py 2>&1 | grep "k2_form 4 zsplit\|k2_form 0 zsplit 0\|cos_x"
  K2FORMS="4" FORMS="0" STREAMS="3 1" WGS="256" ZS="2 3 5" bash $R/tools/k1sweep.sh
done
cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
