#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for q in 4 8 16; do for ns in 3 4 5 6; do echo "GPU_MAX_HW_QUEUES=$q streams=$ns"; GPU_MAX_HW_QUEUES=$q DC_STREAMS=$ns VARIANTS="" PASSES=2 STEPS=200 timeout 200 python tools/r06_quick.py 2>&1 | grep pass; done; done
