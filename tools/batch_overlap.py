"""tools/batch_overlap.py -- how consecutive calls of the batch entry point overlap on the device, per PLACEMENT of the context's streams
(round 6: the same arenas run 37 or 50 us / frame depending on which hardware queues the runtime gave the context and the caller streams).
A -DDC_BT_PROF=1 build (python tools/mkvariant.py BTPROF "-DDC_BT_PROF=1" dense_batch.hip, copied over liblink_amd.so on the GPU box):
every K1 / K2 item leaves 100 MHz timestamps, every K2 workgroup the time it became resident.  TRIALS contexts one after the other on
the same two arena sets; per context: us / frame of 40 calls (timers on), then NC recorded calls alternating the sets -- per call when its
pre_mix role and its gather role started and ended, relative to the first recorded call.
    B=24 TRIALS=5 python tools/batch_overlap.py        -> gpurun_out/batch_overlap.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform

N, C = 100000, 64
B, TRIALS, NC = int(os.environ.get("B", 24)), int(os.environ.get("TRIALS", 5)), int(os.environ.get("NC", 6))
MODE = os.environ.get("MODE", "streams")                # "streams": run() alternating two caller streams; "submit": submit(s + 1) before join(s), one stream
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
NF = 6
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NF)]
bind = lambda o: o.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
sets = [bind(la.ElkCoreBatch(B, N, C, "cos", C // 2, 3, 7, bounds, dev))]
sets.append(bind(la.ElkCoreBatch(B, N, C, "cos", C // 2, 3, 7, bounds, dev, share=sets[0])))
bf = [[frames[(i + 7 * j) % NF][0] for i in range(B)] for j in range(2)]
bc = [[frames[(i + 7 * j) % NF][1] for i in range(B)] for j in range(2)]
CAP = 1 << 16
d1 = [torch.zeros((CAP, 8), dtype=torch.int64, device=dev) for _ in range(NC)]
d2 = [torch.zeros((CAP, 8), dtype=torch.int64, device=dev) for _ in range(NC)]
lib = L.lib()
out = []
mark = torch.zeros(64, device=dev)


class Issuer:
    """issues call s on arena set s % 2: two caller streams (run), or one stream with the join of call s - 1 behind the submit of s"""
    def __init__(self, streams):
        self.streams, self.prev = streams, None

    def call(self, s):
        if os.environ.get("MARK_CALLER"):                 # a tiny kernel on the caller's stream: its hardware queue shows up in a rocprofv3 trace
            with torch.cuda.stream(self.streams[0]):
                mark.fill_(1.0)
        if MODE == "submit":
            _, tk = sets[s % 2].submit(bf[s % 2], bc[s % 2], stream=self.streams[0].cuda_stream)
            if self.prev is not None:
                sets[0].join(self.prev, stream=self.streams[0].cuda_stream)
            self.prev = tk
        else:
            sets[s % 2].run(bf[s % 2], bc[s % 2], stream=self.streams[s % 2].cuda_stream)

    def drain(self):
        if MODE == "submit" and self.prev is not None:
            sets[0].join(self.prev, stream=self.streams[0].cuda_stream)
            self.prev = None
        torch.cuda.synchronize()


for t in range(TRIALS):
    if t:
        sets[0].new_context()
        sets[1].adopt_context(sets[0])
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    prof = lib.link_dc_batch_set_debug(sets[0]._ctx, None, None) == 0
    if not prof and not os.environ.get("RATES_ONLY"):
        raise SystemExit("the library was not built with -DDC_BT_PROF=1 (RATES_ONLY=1: the rates alone, e.g. under rocprofv3)")
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iss = Issuer(streams)
        for s in range(40):
            iss.call(s)
        iss.drain()
        rate = 1e6 * (time.perf_counter() - t0) / (40 * B)
    if not prof:
        iss = Issuer(streams)
        hs = []
        for s in range(12):
            h0 = time.perf_counter()
            iss.call(s)
            hs.append(round(1e6 * (time.perf_counter() - h0)))
        iss.drain()
        print(f"[{MODE}] trial {t}: {rate:.2f} us/frame; host us per call issue (12 calls from idle): {hs}", flush=True)
        continue
    for a, b in zip(d1, d2):
        a.zero_(); b.zero_()
        a[0, 1] = CAP - 1; b[0, 1] = CAP - 1
    torch.cuda.synchronize()
    iss = Issuer(streams)
    for s in range(4):                                    # the recorded calls follow calls in flight, as in the steady state
        iss.call(s)
    for i in range(NC):
        lib.link_dc_batch_set_debug(sets[0]._ctx, d1[i].data_ptr(), d2[i].data_ptr())
        iss.call(i)
    lib.link_dc_batch_set_debug(sets[0]._ctx, None, None)
    iss.drain()
    sets[0].check()
    calls = []
    for i in range(NC):
        k1, k2 = d1[i].cpu().numpy()[1:], d2[i].cpu().numpy()[1:]
        k1, k2 = k1[k1[:, 6] == 1], k2[k2[:, 6] == 1]
        res = k2[k2[:, 0] == 9999][:, 2]
        k2 = k2[k2[:, 0] != 9999]
        calls.append({"k1_start": int(k1[:, 2].min()), "k1_end": int(k1[:, 4].max()), "k2_resident_p50": float(np.median(res)) if len(res) else None,
                      "k2_first_item": int(k2[:, 2].min()), "k2_end": int(k2[:, 4].max()),
                      "k1_busy": float((k1[:, 4] - k1[:, 2]).sum()) / max(1.0, float((k1[:, 4].max() - k1[:, 2].min()) * len(np.unique(k1[:, 5])))),
                      "k2_busy": float((k2[:, 4] - k2[:, 2]).sum()) / max(1.0, float((k2[:, 4].max() - k2[:, 2].min()) * len(np.unique(k2[:, 5]))))})
    T0 = calls[0]["k1_start"]
    u = lambda x: None if x is None else round((x - T0) / 100.0, 1)
    rec = {"trial": t, "us_per_frame": round(rate, 2),
           "calls": [{"call": i, "k1": [u(c["k1_start"]), u(c["k1_end"])], "k2_resident_p50": u(c["k2_resident_p50"]), "k2": [u(c["k2_first_item"]), u(c["k2_end"])],
                      "k1_busy": round(c["k1_busy"], 3), "k2_busy": round(c["k2_busy"], 3)} for i, c in enumerate(calls)]}
    rec["call_period_us"] = round((calls[-1]["k2_end"] - calls[0]["k2_end"]) / 100.0 / (NC - 1), 1)
    rec["k1_gap_between_calls_us"] = [u(calls[i + 1]["k1_start"]) - u(calls[i]["k1_end"]) for i in range(NC - 1)]
    rec["k2_gap_between_calls_us"] = [u(calls[i + 1]["k2_first_item"]) - u(calls[i]["k2_end"]) for i in range(NC - 1)]
    out.append(rec)
    print(f"[{MODE}] trial {t} queue delays a>b a>c b>c st>a st>b st>c | b>a c>a c>b a>st b>st c>st: {sets[0].probe_streams(streams[0].cuda_stream)}")
    print(f"[{MODE}] trial {t}: {rate:.2f} us/frame; call period {rec['call_period_us']} us = {rec['call_period_us'] / B:.2f} us/frame")
    for c in rec["calls"]:
        print(f"   call {c['call']}: pre_mix {c['k1'][0]:8.1f} .. {c['k1'][1]:8.1f} (busy {c['k1_busy']:.2f})   gather resident {c['k2_resident_p50']:8.1f}, items {c['k2'][0]:8.1f} .. {c['k2'][1]:8.1f} (busy {c['k2_busy']:.2f})")
    print(f"   pre_mix end -> next call's pre_mix start: {[round(x, 1) for x in rec['k1_gap_between_calls_us']]};  gather end -> next call's first gather item: {[round(x, 1) for x in rec['k2_gap_between_calls_us']]}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/batch_overlap_{MODE}.json", "w"), indent=1)
