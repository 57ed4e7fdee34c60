#!/bin/bash
# tools/ab_variants.sh -- A/B of prebuilt library variants (link_amd/lib/variants/lib_<X>.so, built in the container with
# different -D flags): each variant is copied over liblink_amd.so on the (scratch) GPU box, then SCRIPT runs under
# rocprofv3 and the kernels matching PAT are listed.   VARIANTS="A B" SCRIPT=tools/x.py PAT=centre_sum bash tools/ab_variants.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/link_amd/lib/liblink_amd.so /tmp/lib_orig.so
for v in ${VARIANTS:-A B}; do
  cp $R/link_amd/lib/variants/lib_$v.so $R/link_amd/lib/liblink_amd.so
  for s in ${SCRIPTS:-$SCRIPT}; do
    OUT=/tmp/ab_$v; rm -rf $OUT; mkdir -p $OUT
    timeout ${TMO:-150} rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/$s ${ARGS:-} > $OUT/run.log 2>&1
    db=$(find $OUT -name "*.db" | head -1)
    echo "== variant $v  $s"
    [ -n "$db" ] && python $R/tools/rocpd_stats.py $db | grep -E "${PAT:-.}" | cut -c1-${COLS:-150}
  done
done
cp /tmp/lib_orig.so $R/link_amd/lib/liblink_amd.so
