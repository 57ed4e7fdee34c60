R=$GRAFT_REPO_ROOT
for p1 in 0 2048 4096; do for p2 in 0 2048; do
  LINK_BENCH_K1_PAD=$p1 LINK_BENCH_K2_PAD=$p2 timeout 300 python $R/bench.py --steps 300 --warmup 30 --streams 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('k1_pad $p1 k2_pad $p2: %.2f us/frame' % d['us_per_frame'])
"
done; done
