#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
for z in 2 1 3; do echo "zsplit=$z"; LINK_DC_BATCH_ZSPLIT=$z B=32 SETS=2 PASSES=2 timeout 300 python tools/batch_bench.py 2>&1 | grep -E "pass|equal"; done
