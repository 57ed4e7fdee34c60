#!/bin/bash
# tools/ab_lib.sh -- one script under prebuilt library variants on ONE box, alternating:  LIBS="EA0 EA2" SCRIPT=tools/centre_ab.py bash tools/ab_lib.sh
# ("" = the built library).  Variants: python tools/mkvariant.py <NAME> "-D..." [file.hip ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
for pass in 1 2; do
  for v in default $LIBS; do
    if [ "$v" = default ]; then cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so; else cp link_amd/lib/variants/lib_$v.so link_amd/lib/liblink_amd.so; fi
    echo "== pass $pass variant $v"
    VARIANTS="$SCRIPT_VARIANTS" timeout ${TMO:-300} python $SCRIPT 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-4}
  done
done
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
