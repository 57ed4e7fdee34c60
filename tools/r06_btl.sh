#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2
for b in 8 24 32; do echo "B=$b"; B=$b SETS=2 PASSES=2 timeout 300 python tools/batch_bench.py 2>&1 | grep pass; done
echo "B=24 1 set"; B=24 SETS=1 PASSES=2 timeout 300 python tools/batch_bench.py 2>&1 | grep pass
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_BTPROF.so link_amd/lib/liblink_amd.so
B=24 timeout 300 python tools/batch_timeline.py > $O/batch_timeline.txt 2>&1
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
grep -E "^K1 frame|^K2 frame|us_per_frame" $O/batch_timeline.txt
