"""tools/batch_timeline.py -- where the resident workgroups of the persistent batch kernels spend a call (round 6): every K1 item
(frame, range) and K2 item (frame, tile) leaves a row of 100 MHz timestamps (a -DDC_BT_PROF=1 build of csrc/dense_batch.hip:
python tools/mkvariant.py BTPROF "-DDC_BT_PROF=1" dense_batch.hip, copied over liblink_amd.so on the GPU box).
Prints per role: items, mean / p50 / p90 / max item time, time between a wave's consecutive items (control + waits), the share of the
call each role's slots were busy, and per frame when its K1 items / K2 items started and ended (who waits for whom).
    B=24 python tools/batch_timeline.py        -> gpurun_out/batch_timeline.json"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform

N, C, B = 100000, 64, int(os.environ.get("B", 24))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
NF = 6
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NF)]
batch = la.ElkCoreBatch(B, N, C, "cos", C // 2, 3, 7, bounds, dev)
batch.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
bf, bc = [frames[i % NF][0] for i in range(B)], [frames[i % NF][1] for i in range(B)]
for _ in range(5):
    batch.run(bf, bc)
torch.cuda.synchronize()
CAP = 1 << 17
d1 = torch.zeros((CAP, 8), dtype=torch.int64, device=dev)
d2 = torch.zeros((CAP, 8), dtype=torch.int64, device=dev)
d1[0, 1] = CAP - 1
d2[0, 1] = CAP - 1
rc = L.lib().link_dc_batch_set_debug(batch._ctx, d1.data_ptr(), d2.data_ptr())
if rc != 0:
    raise SystemExit("the library was not built with -DDC_BT_PROF=1 (python tools/mkvariant.py BTPROF \"-DDC_BT_PROF=1\" dense_batch.hip)")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
batch.run(bf, bc)
e1.record()
torch.cuda.synchronize()
batch.check()
L.lib().link_dc_batch_set_debug(batch._ctx, None, None)
wall = 1e3 * e0.elapsed_time(e1)
k1 = d1.cpu().numpy(); k2 = d2.cpu().numpy()
k1, k2 = k1[1:], k2[1:]
k1, k2 = k1[k1[:, 6] == 1], k2[k2[:, 6] == 1]                      # rows written (every item has its own place)
entry2 = k2[k2[:, 0] == 9999]                                      # K2 workgroups' residency stamps
k2 = k2[k2[:, 0] != 9999]
T0 = min(k1[:, 2].min(), k2[:, 2].min())
us = lambda t: (t - T0) / 100.0


def role(rows, who_col, name):
    dur = (rows[:, 4] - rows[:, 2]) / 100.0
    pre = (rows[:, 3] - rows[:, 2]) / 100.0
    out = {"items": int(len(rows)), "item_us_mean": float(dur.mean()), "item_us_p50": float(np.median(dur)), "item_us_p90": float(np.percentile(dur, 90)),
           "item_us_max": float(dur.max())}
    gaps = []
    busy = 0.0
    for w in np.unique(rows[:, who_col]):
        r = rows[rows[:, who_col] == w]
        r = r[np.argsort(r[:, 2])]
        gaps.extend(((r[1:, 2] - r[:-1, 4]) / 100.0).tolist())
        busy += float(((r[:, 4] - r[:, 2]) / 100.0).sum())
    span = (rows[:, 4].max() - rows[:, 2].min()) / 100.0
    out.update({"workers": int(len(np.unique(rows[:, who_col]))), "gap_between_items_us_mean": float(np.mean(gaps)) if gaps else None,
                "gap_between_items_us_p90": float(np.percentile(gaps, 90)) if gaps else None, "role_span_us": float(span),
                "busy_share_of_span": busy / (span * len(np.unique(rows[:, who_col])))})
    if name == "k1":
        out["first_loads_wait_us_mean"] = float(pre.mean())
    else:
        out["tile_us_mean"] = float(pre.mean())                   # start -> mapper through (the tile itself)
        out["next_item_control_us_mean"] = float(((rows[:, 4] - rows[:, 3]) / 100.0).mean())
    return out


def spread(rows, who_col):
    cnt = np.array([int((rows[:, who_col] == w).sum()) for w in np.unique(rows[:, who_col])])
    first = np.array([float(us(rows[rows[:, who_col] == w][:, 2].min())) for w in np.unique(rows[:, who_col])])
    return {"items_per_worker_min_mean_max": [int(cnt.min()), float(cnt.mean()), int(cnt.max())],
            "first_item_start_us_p50_p90_max": [float(np.median(first)), float(np.percentile(first, 90)), float(first.max())]}


summary = {"frames": B, "wall_us": wall, "us_per_frame": wall / B, "k1": role(k1, 5, "k1"), "k2": role(k2, 5, "k2")}
per_frame = []
for f in range(B):
    a, b = k1[k1[:, 0] == f], k2[k2[:, 0] == f]
    per_frame.append({"frame": f, "k1_first_start": float(us(a[:, 2].min())), "k1_last_end": float(us(a[:, 4].max())), "k1_items": int(len(a)),
                      "k2_first_start": float(us(b[:, 2].min())), "k2_last_end": float(us(b[:, 4].max())), "k2_items": int(len(b))})
summary["k1"].update(spread(k1, 5)); summary["k2"].update(spread(k2, 5))
summary["k2"]["last_end_us_by_xcd_queue"] = [float(us(k2[(k2[:, 1] & 7) == q][:, 4].max())) for q in range(8)]
if len(entry2):
    e = (entry2[:, 2] - T0) / 100.0
    summary["k2"]["workgroup_resident_at_us_min_p50_max"] = [float(e.min()), float(np.median(e)), float(e.max())]
print(json.dumps(summary, indent=1))
print("frame  K1 first..last end   K2 first..last end   (us from the first item's start)")
for r in per_frame:
    print(f"{r['frame']:5d}  {r['k1_first_start']:8.1f} {r['k1_last_end']:8.1f}   {r['k2_first_start']:8.1f} {r['k2_last_end']:8.1f}")
d = np.diff([r["k1_last_end"] for r in per_frame]); print("K1 frame period (last-end to last-end) mean %.2f us" % d.mean())
d = np.diff([r["k2_last_end"] for r in per_frame]); print("K2 frame period mean %.2f us" % d.mean())
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"summary": summary, "per_frame": per_frame}, open("gpurun_out/batch_timeline.json", "w"), indent=1)
np.savez_compressed("gpurun_out/batch_timeline_rows.npz", k1=k1, k2=k2, entry2=entry2, t0=np.array([T0]))
