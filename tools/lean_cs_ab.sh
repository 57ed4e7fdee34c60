# tools/lean_cs_ab.sh -- lean form A/B on the LiDAR stage frames: CS=0|1 (channel-split first launch), STAGES="2 3 6 7"
pr() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  $1', d['stage'], d['n'], d['c'], 'cold', d.get('lean_cold_us'), 'warm', d.get('lean_warm_us'), 'bitwise', d.get('lean_repeat_bitwise'))
"; }
for rep in 1 2; do
for cs in ${CSV:-0 1}; do for st in ${STAGES:-2 3 6 7}; do CS=$cs STAGE=$st FORM=lean timeout 100 python tools/lidar_core.py 2>/dev/null | pr "cs=$cs"; done; done
done
