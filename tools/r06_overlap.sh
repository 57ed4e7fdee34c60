#!/bin/bash
cd $GRAFT_REPO_ROOT
cp link_amd/lib/liblink_amd.so /tmp/lib_orig.so
cp link_amd/lib/variants/lib_BTPROF.so link_amd/lib/liblink_amd.so
for m in streams submit; do MODE=$m GPU_MAX_HW_QUEUES=8 B=${B:-24} TRIALS=${TRIALS:-5} timeout 400 python tools/batch_overlap.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/batch_overlap.txt
cp /tmp/lib_orig.so link_amd/lib/liblink_amd.so
cat gpurun_out/batch_overlap.txt
