#!/bin/bash
# tools/r06_batch.sh -- the batch entry point on the GPU: parity tests (hard timeout: a persistent kernel that waits for ever must
# not hold the box), then the A/B against three plans on three streams
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/r06_batch_tests.txt 2>&1; echo "tests rc=$?" >> $O/r06_batch_tests.txt
tail -6 $O/r06_batch_tests.txt
for rpw in 1 2 4; do
B=24 SETS=2 LINK_DC_BATCH_RPW=$rpw timeout 300 python tools/batch_bench.py > $O/r06_batch_bench_rpw$rpw.txt 2>&1; echo "rc=$?" >> $O/r06_batch_bench_rpw$rpw.txt
echo "rpw $rpw"; tail -5 $O/r06_batch_bench_rpw$rpw.txt
done
B=24 SETS=1 timeout 300 python tools/batch_bench.py > $O/r06_batch_bench_1set.txt 2>&1
B=8 SETS=2 timeout 300 python tools/batch_bench.py > $O/r06_batch_bench_b8.txt 2>&1
tail -3 $O/r06_batch_bench_1set.txt; tail -3 $O/r06_batch_bench_b8.txt
