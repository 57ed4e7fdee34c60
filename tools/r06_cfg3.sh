#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_cfg3; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_train.py tests/test_gpu_encoder.py tests/test_gpu_bench.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python bench.py --workload cfg3 --steps 30 --warmup 3 2>$O/cfg3.err | grep '^{' > $O/bench_cfg3.json; tail -c 300 $O/cfg3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_cfg3/bench_cfg3.json'))
print('fwd ms',d['ms_per_step'],'fwd_bwd_ms',d['fwd_bwd_ms'])
c=d['conv_roofline']; print('conv sum_us',c['sum_us'],'roof',c['sum_roofline_us'],'frac',c['frac'])
for s in c['shapes'][:12]: print(s)
PY
D=/tmp/tr; rm -rf $D; mkdir -p $D
(cd /tmp; K=15 REPS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $GRAFT_REPO_ROOT/tools/cfg3train_prof.py > $GRAFT_REPO_ROOT/$O/train_prof.log 2>&1)
db=$(find $D -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/cfg3_train_kernel_stats.csv | head -14
D=/tmp/cv; rm -rf $D; mkdir -p $D
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 30 --warmup 3 > /dev/null 2>&1)
db=$(find $D -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py $db $O/cfg3_kernel_stats.csv | head -8
