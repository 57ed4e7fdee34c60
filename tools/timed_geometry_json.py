"""tools/timed_geometry_json.py -- profiles/timed_geometry.json: what bench.py's `roofline` object is built from (round 6, VERDICT round 5
"next 2": the per-kernel figures of the line must describe the TIMED launch geometry and follow from the committed rocprof files).

    python tools/timed_geometry_json.py <kernel_stats_streams3.csv> <pmc_counters_streams3.txt> <out.json> <commit> <label prefix>

Per kernel of the dense-cell step as the timed region launches it (three plans on three streams, frames_in_flight = 3: the cell-range
pre_mix kernel at 256 workgroups, the quad gather kernel with 2 z-segments, the slot insert): rocprofv3 --kernel-trace average duration
of the bench command itself, launches, and the memory-side bytes per launch from the separate --pmc passes over tools/dcstep3.py in the
same geometry (2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: KiB counters, the gfx950 wide-read correction of MI355X_MICROARCH.md)."""
import ast
import csv
import json
import re
import sys

stats, pmc, out, commit, label = sys.argv[1:6]
KEYS = (("index", "k_dc_index"), ("premix_modsum", "k_dc_premix_modsumILi64"), ("gather_demod", "k_dc_gather_demod_quad"))
kern = {}
for row in csv.DictReader(open(stats)):
    for key, pat in KEYS:
        if pat in row["name"] and "_mm" not in row["name"] and key not in kern:
            kern[key] = {"rocprof_name": row["name"][:96], "launches": int(row["calls"]), "avg_us": float(row["avg_us"]),
                         "min_us": float(row["min_us"]), "max_us": float(row["max_us"])}
vals = {}
for line in open(pmc):
    m = re.match(r"^(\S.*?) (\{.*\}) n=\d+", line.strip())
    if not m:
        continue
    name, d = m.group(1), ast.literal_eval(m.group(2))
    key = "index" if "k_dc_index" in name else "premix_modsum" if "premix_modsum" in name else "gather_demod" if "gather_demod" in name else None
    if key:
        vals.setdefault(key, {}).update(d)
for key, v in vals.items():
    if key in kern and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        kern[key]["traffic_bytes_per_launch"] = int(round(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024))
    if key in kern and v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        kern[key]["mfma_busy_cycles_per_launch"] = int(round(v["SQ_VALU_MFMA_BUSY_CYCLES"]))
# the batch entry point's three kernels in the same bench command (its second timed path: calls of 48 frames): rocprofv3 averages per
# launch = per CALL; the averages include the launches of the calibration's other contexts and of the rows check
BKEYS = (("insert", "k_dc_batch_insert"), ("premix_modsum", "k_dc_batch_k1ILi0"), ("gather_demod", "k_dc_batch_k2ILi0ELi3"))
bkern = {}
for row in csv.DictReader(open(stats)):
    for key, pat in BKEYS:
        if pat in row["name"] and key not in bkern:
            bkern[key] = {"rocprof_name": row["name"][:96], "launches": int(row["calls"]), "avg_us": float(row["avg_us"]),
                          "min_us": float(row["min_us"]), "max_us": float(row["max_us"])}
batch = ({"frames_per_call": int(sys.argv[6]) if len(sys.argv) > 6 else 48,
          "geometry": "link_dc_batch_submit / link_dc_batch_join: calls of 48 frames, two arena sets, two calls in flight (cfg2: N = 100000, C = 64, cos, r = 3, s = 7)",
          "kernels": bkern} if len(bkern) == 3 else None)
json.dump({"commit": commit, "batch": batch, "geometry": "three ElkCorePlan on three HIP streams, frames_in_flight = 3 (cfg2: N = 100000, C = 64, cos, r = 3, s = 7)",
           "source_kernel_stats": f"{label}_kernel_stats_streams3.csv (rocprofv3 --kernel-trace --stats of `python bench.py --steps 100 --warmup 10 --streams 3`)",
           "source_pmc": f"{label}_pmc_counters_streams3.txt (rocprofv3 --pmc, one counter set per pass, over tools/dcstep3.py DC_STREAMS=3)",
           "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps(kern, indent=1))
