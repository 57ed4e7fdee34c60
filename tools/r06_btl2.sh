#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_BTPROF.so link_amd/lib/liblink_amd.so
B=24 timeout 300 python tools/batch_timeline.py > $O/batch_timeline.txt 2>&1
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
grep -E "workgroup_resident" -A4 $O/batch_timeline.txt; grep "^ *[0-9]* " $O/batch_timeline.txt | head -5
