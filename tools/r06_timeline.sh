#!/bin/bash
# tools/r06_timeline.sh -- where the K1 / K2 slots idle: -DDC_PROF=2 build (timers on the 100 MHz clock all XCDs share), timed geometry
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_PROF2.so link_amd/lib/liblink_amd.so
timeout 300 python tools/slot_timeline.py > $O/slot_timeline_default.txt 2>&1; cp $O/slot_timeline.json $O/slot_timeline_default.json
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
# the refactored library (dense_k1_impl.h split) against the round-5 build, same box
for rep in 1 2; do
for v in orig R5; do
  [ $v = orig ] && cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so || cp link_amd/lib/variants/lib_$v.so link_amd/lib/liblink_amd.so
  echo "== $v" >> $O/r06_ab_refactor.txt
  VARIANTS="" PASSES=2 timeout 300 python tools/r06_quick.py >> $O/r06_ab_refactor.txt 2>&1
done
done
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
tail -40 $O/slot_timeline_default.txt; grep -E "==|pass" $O/r06_ab_refactor.txt
