"""tools/slot_timeline.py -- WHERE the resident workgroups idle (round 6): per-wave start / end timestamps of every pre_mix (K1) and
gather (K2) launch of the TIMED geometry (three plans, three streams, frames_in_flight = 3), from a -DDC_PROF=1 build of the library
(python tools/mkvariant.py PROF "-DDC_PROF=1"; copy lib_PROF.so over liblink_amd.so on the GPU box).

A CU holds at most one K1 workgroup (4 waves, one per SIMD) and one K2 workgroup (8 waves) of the frames in flight (LDS: 81 + 79 KB).
The question this answers: what share of the 1 024 K1 wave slots and of the 256 K2 workgroup slots is occupied over a steady-state
window, and how long one K1 wave / one K2 workgroup lives -- i.e. whether a persistent, queue-fed arrangement has idle slots to
recover, and which role is the bottleneck.  Output: a summary on stdout, gpurun_out/slot_timeline.json.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from bench import s_uniform

N, C, NS = 100000, 64, int(os.environ.get("DC_STREAMS", 3))
REC = int(os.environ.get("REC", 8))                    # recorded steps (x NS frames)
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
tune = {k: int(os.environ[e]) for k, e in (("k1_wgs", "DC_K1_WGS"), ("k2_zsplit", "DC_K2_ZSPLIT"),
                                           ("k1_lds_pad", "DC_K1_PAD")) if os.environ.get(e) not in (None, "")}
frames, plans, streams = [], [], []
for k in range(NS):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS, **tune)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
           blk.norm.weight, blk.norm.bias)
    plans.append(p)
    streams.append(torch.cuda.Stream(device=dev))
K1W, K2W = 4096, 512 * 8                               # dbg rows per launch (waves): upper bounds
k1dbg = torch.zeros((REC, NS, K1W, 8), dtype=torch.int64, device=dev)
k2dbg = torch.zeros((REC, NS, K2W, 8), dtype=torch.int64, device=dev)


def run(steps, rec=False):
    for s in range(steps):
        for j in range(NS):
            if rec:
                plans[j].buf.tune.k1_dbg = k1dbg[s, j].data_ptr()
                plans[j].buf.tune.k2_dbg = k2dbg[s, j].data_ptr()
            with torch.cuda.stream(streams[j]):
                plans[j].run(*frames[j])


run(60)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run(REC, rec=True)
for s_ in streams:
    torch.cuda.current_stream().wait_stream(s_)
e1.record()
torch.cuda.synchronize()
wall_us = 1e3 * e0.elapsed_time(e1)
k1 = k1dbg.cpu().numpy()
k2 = k2dbg.cpu().numpy()
if not (k1[..., 5] > 0).any():
    raise SystemExit("no timer data: the library was not built with -DDC_PROF=1 (tools/mkvariant.py PROF \"-DDC_PROF=1\")")

# intervals: K1 per wave (start = [7], duration = [5]); K2 per workgroup from its producer wave 0 rows (kind [7] == 1: start [4], duration [0])
iv1, iv2, per_launch = [], [], []
for s in range(REC):
    for j in range(NS):
        a = k1[s, j]
        a = a[a[:, 5] > 0]
        b = k2[s, j]
        bp = b[(b[:, 7] == 1) & (b[:, 0] > 0)]
        bc = b[(b[:, 7] == 2) & (b[:, 0] > 0)]
        iv1.append(np.stack([a[:, 7], a[:, 7] + a[:, 5]], 1))
        # one row per workgroup: earliest producer start, latest end over its waves (producers and consumers)
        wg_start = bp[:, 4].reshape(-1, 4).min(1) if len(bp) % 4 == 0 and len(bp) else bp[:, 4]
        wg_end = (bp[:, 4] + bp[:, 0]).reshape(-1, 4).max(1) if len(bp) % 4 == 0 and len(bp) else bp[:, 4] + bp[:, 0]
        iv2.append(np.stack([wg_start, wg_end], 1))
        per_launch.append({"step": s, "stream": j, "k1_waves": int(len(a)), "k1_first": int(a[:, 7].min()), "k1_last_end": int((a[:, 7] + a[:, 5]).max()),
                           "k1_wave_ticks_mean": float(a[:, 5].mean()), "k1_wave_ticks_p50": float(np.median(a[:, 5])), "k1_wave_ticks_max": int(a[:, 5].max()),
                           "k1_phase_ticks_mean": {n: float(a[:, i].mean()) for i, n in enumerate(("staging", "cell_section", "fill", "tile_body", "cell_sums"))},
                           "k1_tiles_per_wave": float(a[:, 6].mean()),
                           "k2_wgs": int(len(wg_start)), "k2_first": int(wg_start.min()), "k2_last_end": int(wg_end.max()),
                           "k2_wg_ticks_mean": float((wg_end - wg_start).mean()), "k2_wg_ticks_max": int((wg_end - wg_start).max()),
                           "k2_prod_phase_ticks_mean": {"dma_wait": float(bp[:, 1].mean()), "barrier": float(bp[:, 2].mean()), "box_sums": float(bp[:, 3].mean())},
                           "k2_cons_phase_ticks_mean": {"barrier": float(bc[:, 2].mean()), "rounds_work": float(bc[:, 4].mean())} if len(bc) else None})
iv1, iv2 = np.concatenate(iv1), np.concatenate(iv2)
t_lo, t_hi = min(iv1[:, 0].min(), iv2[:, 0].min()), max(iv1[:, 1].max(), iv2[:, 1].max())
ticks_per_us = (t_hi - t_lo) / wall_us
# steady-state window: from the first K1 start of recorded step 1 to the first K1 start of the last recorded step
w_lo = min(p["k1_first"] for p in per_launch if p["step"] == 1)
w_hi = min(p["k1_first"] for p in per_launch if p["step"] == REC - 1)
ts = np.linspace(w_lo, w_hi, 4000)


def active(iv, t):
    return ((iv[:, 0][None, :] <= t[:, None]) & (iv[:, 1][None, :] > t[:, None])).sum(1)


a1, a2 = active(iv1, ts), active(iv2, ts)
frames_in_window = (REC - 2) * NS
us_per_frame = (w_hi - w_lo) / ticks_per_us / frames_in_window
tk = lambda x: x / ticks_per_us
summary = {
    "ticks_per_us": ticks_per_us, "wall_us_recorded": wall_us, "us_per_frame_in_window": us_per_frame,
    "k1_wave_slots": 1024, "k1_waves_active_mean": float(a1.mean()), "k1_waves_active_p10_p50_p90": [float(np.percentile(a1, q)) for q in (10, 50, 90)],
    "k1_slot_occupancy": float(a1.mean() / 1024),
    "k2_wg_slots_beside_k1": 256, "k2_wgs_active_mean": float(a2.mean()), "k2_wgs_active_p10_p50_p90": [float(np.percentile(a2, q)) for q in (10, 50, 90)],
    "k2_slot_occupancy_of_256": float(a2.mean() / 256),
    "k1_wave_us_mean": tk(np.mean([p["k1_wave_ticks_mean"] for p in per_launch])), "k1_wave_us_max": tk(np.mean([p["k1_wave_ticks_max"] for p in per_launch])),
    "k1_launch_span_us_mean": tk(np.mean([p["k1_last_end"] - p["k1_first"] for p in per_launch])),
    "k2_wg_us_mean": tk(np.mean([p["k2_wg_ticks_mean"] for p in per_launch])), "k2_wg_us_max": tk(np.mean([p["k2_wg_ticks_max"] for p in per_launch])),
    "k2_launch_span_us_mean": tk(np.mean([p["k2_last_end"] - p["k2_first"] for p in per_launch])),
    "k1_phase_us_mean": {n: tk(np.mean([p["k1_phase_ticks_mean"][n] for p in per_launch])) for n in ("staging", "cell_section", "fill", "tile_body", "cell_sums")},
    "k1_tiles_per_wave": float(np.mean([p["k1_tiles_per_wave"] for p in per_launch])),
    "k2_prod_phase_us_mean": {n: tk(np.mean([p["k2_prod_phase_ticks_mean"][n] for p in per_launch])) for n in ("dma_wait", "barrier", "box_sums")},
    "tuning": tune,
    "note": "timers compiled in (-DDC_PROF=1) cost a few percent; occupancies are shares of the slots ONE K1 + ONE K2 workgroup per CU give",
}
print(json.dumps(summary, indent=1))
# gap between a frame's K1 end and its K2 start, and between insert end (= K1 first start, approx.) -- the stream-ordered launch boundaries
gaps = [tk(p["k2_first"] - p["k1_last_end"]) for p in per_launch]
print("K1 last end -> K2 first start per frame (us): mean %.2f  min %.2f  max %.2f" % (np.mean(gaps), np.min(gaps), np.max(gaps)))
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"summary": summary, "per_launch": per_launch, "gaps_k1_to_k2_us": gaps}, open("gpurun_out/slot_timeline.json", "w"), indent=1)
