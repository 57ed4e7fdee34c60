R=$GRAFT_REPO_ROOT
run() { timeout 120 python $R/bench.py --steps 200 --warmup 20 --streams $1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   $2 streams $1: %.2f us/frame' % d['us_per_frame'])
"; }
run 3 default; run 3 default
for k2 in 8 4 2; do LINK_BENCH_K2_FORM=$k2 run 3 "k2_form=$k2"; done
for k2 in 0 8 4; do LINK_BENCH_K1_FORM=2 LINK_BENCH_K1_WGS=1024 LINK_BENCH_K2_FORM=$k2 run 3 "k1mm k2_form=$k2"; done
for k2 in 0 4; do LINK_BENCH_K2_FORM=$k2 run 4 "k2_form=$k2"; done
LINK_BENCH_K2_ZSPLIT=1 run 3 "zsplit1"; LINK_BENCH_K2_ZSPLIT=3 run 3 "zsplit3"
run 3 default
