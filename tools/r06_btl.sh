#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
B=24 SETS=2 timeout 300 python tools/batch_bench.py 2>&1 | tail -5
echo "GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 B=24 SETS=2 timeout 300 python tools/batch_bench.py 2>&1 | tail -4
B=24 SETS=1 timeout 300 python tools/batch_bench.py 2>&1 | tail -3
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_BTPROF.so link_amd/lib/liblink_amd.so
B=24 timeout 300 python tools/batch_timeline.py > $O/batch_timeline.txt 2>&1
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
cat $O/batch_timeline.txt | grep -v "^ \|^{\|^}" 
grep -E "us_per_frame|item_us_mean|busy_share|first_item_start" -A0 $O/batch_timeline.txt
