#!/bin/bash
# tools/gpu_check.sh -- run the GPU test files one by one under hard timeouts (a hung kernel must not
# eat the gpurun budget), then smoke + bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
T=${T:-150}
for f in ${FILES:-tests/test_gpu_aggregate.py tests/test_gpu_batch.py tests/test_gpu_bench.py tests/test_gpu_block_driver.py tests/test_gpu_conv.py tests/test_gpu_core_lidar.py tests/test_gpu_dense.py tests/test_gpu_detstage.py tests/test_gpu_elk.py tests/test_gpu_elk_tiles.py tests/test_gpu_encoder.py tests/test_gpu_index.py tests/test_gpu_index_first.py tests/test_gpu_lean.py tests/test_gpu_ops.py tests/test_gpu_pointvoxel.py tests/test_gpu_split_paths.py tests/test_gpu_train.py}; do
  b=$(basename $f .py)
  timeout $T python -m pytest $f -m gpu -x -q --timeout=60 --timeout-method=thread > gpurun_out/$b.log 2>&1
  echo "== $f rc=$? : $(tail -1 gpurun_out/$b.log)"
  grep -E "FAILED|Error|Timeout|timeout" gpurun_out/$b.log | head -8
done
if [ -z "$NOSMOKE" ]; then
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke rc=$? : $(tail -1 gpurun_out/smoke.log)"
  timeout 200 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2>&1; echo "== bench rc=$?"; tail -3 gpurun_out/bench.log
fi
