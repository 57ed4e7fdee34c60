#!/bin/bash
# tools/r06_first.sh -- round 6, first GPU call: where the K1 / K2 slots idle (PROF build), k1_pipe A/B in the timed geometry,
# the new parity tests (r=3 compiled-reference pin, LiDAR seeds).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O; rm -f $O/lidar_parity_maxima.jsonl
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_PROF.so link_amd/lib/liblink_amd.so
timeout 300 python tools/slot_timeline.py > $O/slot_timeline_default.txt 2>&1; cp $O/slot_timeline.json $O/slot_timeline_default.json
DC_K1_PIPE=1 timeout 300 python tools/slot_timeline.py > $O/slot_timeline_pipe.txt 2>&1; cp $O/slot_timeline.json $O/slot_timeline_pipe.json
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
VARIANTS=";k1_pipe=1;k1_pipe=1,k1_lds_pad=0;k2_zsplit=1" timeout 400 python tools/r06_quick.py > $O/r06_quick.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_core_lidar.py tests/test_gpu_aggregate.py tests/test_gpu_elk.py -x -q -m gpu > $O/r06_tests1.txt 2>&1
tail -5 $O/slot_timeline_default.txt; tail -3 $O/r06_quick.txt; tail -3 $O/r06_tests1.txt
