# tools/lean_pm_ab.sh -- lean form A/B on the LiDAR stage frames: PM=0 (scratch matrix X between launches 1 and 2) against PM=1
# (launch 1 = slot insert alone, pre_mix inside launch 2); STAGES="0 1 4 5 6 3"
for pm in ${PMV:-0 1}; do for st in ${STAGES:-0 1 4 5 6 3 2}; do PM=$pm STAGE=$st FORM=lean timeout 100 python tools/lidar_core.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  pm=$pm', d['stage'], d['n'], d['c'], 'cold', d.get('lean_cold_us'), 'warm', d.get('lean_warm_us'), d.get('lean_repeat_bitwise'), d.get('lean_err'))
"; done; done
