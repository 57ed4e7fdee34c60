"""Kernel timeline of ELKBlock.forward on new coordinate sets (cfg2) from a rocprofv3 database, or -- run as the traced
script -- the loop itself.  TAG=blk SCRIPT=tools/block_timeline.py bash tools/profile_cmd.sh, then
python tools/block_timeline.py <trace.db>   (profile_cmd.sh removes the db: this script is also called from the wrapper below)"""
import sys, time
if len(sys.argv) > 1 and sys.argv[1].endswith(".db"):
    import sqlite3
    con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
    syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select kernel_id, start, end, {qcol or 0} from rocpd_kernel_dispatch order by start").fetchall()
    # the last call: find the last k_dc_index_probe_bbox
    idx = [i for i, r in enumerate(rows) if "k_dc_index_probe_bbox" in syms.get(r[0], "")]
    for which in (idx[-3:-1] if len(idx) >= 3 else idx[-1:]):
        t0 = rows[which][1]
        print("---- call starting at dispatch", which)
        for r in rows[which: which + 16]:
            print(f"  +{(r[1]-t0)/1e3:8.1f} .. +{(r[2]-t0)/1e3:8.1f} us  q{r[3]}  {syms.get(r[0], '?').split('(')[0][:80]}")
    sys.exit(0)
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from link_amd import elk
from helpers import s_uniform
dev = torch.device("cuda:0")
N, C = 100000, 64
coords = s_uniform(N, grid=256, seed=0).to(dev)
feats = torch.randn(N, C, device=dev)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
def cold():
    st = la.SparseTensor(feats, coords, 1)
    with torch.no_grad():
        blk(st, 7, 3)
for _ in range(10): cold()
torch.cuda.synchronize()
lib = elk.L.lib()
f0 = lib.link_elk_block_forward
acc = [0.0, 0]
class W:
    def __call__(self, *a):
        t = time.perf_counter(); r = f0(*a); acc[0] += time.perf_counter() - t; acc[1] += 1; return r
lib.link_elk_block_forward = W()
t0 = time.perf_counter()
for _ in range(100): cold()
torch.cuda.synchronize()
print("cold wall %.1f us; inside link_elk_block_forward %.1f us x %d" % (1e6 * (time.perf_counter() - t0) / 100, 1e6 * acc[0] / max(acc[1], 1), acc[1]))
