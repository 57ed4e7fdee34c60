#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel (our kernels only)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")
        if not name.startswith("k_") and "k_" not in name[:6]:
            if "link" not in name and not name.startswith("void k_"):
                pass
        short = name.split("(")[0].replace("void ", "")[:48]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if not (k.startswith("k_") or "k_" in k):
        continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
