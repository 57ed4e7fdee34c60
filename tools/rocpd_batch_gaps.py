#!/usr/bin/env python
"""tools/rocpd_batch_gaps.py -- from a rocprofv3 rocpd database of a run of batch calls: per (pre_mix queue, gather queue, insert queue)
combination, how long after the previous call's pre_mix kernel ended the next one started, and whether that start followed the previous
gather kernel's end (calls running one after the other).  Usage: rocpd_batch_gaps.py trace.db"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
rows = cur.execute("select kernel_id, start, end, queue_id from rocpd_kernel_dispatch order by start").fetchall()
k1 = [(s, e, q) for k, s, e, q in rows if "k_dc_batch_k1" in syms.get(k, "")]
k2 = [(s, e, q) for k, s, e, q in rows if "k_dc_batch_k2" in syms.get(k, "")]
ins = [(s, e, q) for k, s, e, q in rows if "k_dc_batch_insert" in syms.get(k, "")]
fill = [(s, e, q) for k, s, e, q in rows if "fillBuffer" in syms.get(k, "")]
mark = [(s, e, q) for k, s, e, q in rows if "FillFunctor" in syms.get(k, "")]
agg = collections.defaultdict(list)
for i in range(1, min(len(k1), len(k2), len(ins))):
    if k1[i][0] - k1[i - 1][1] > 5e6:
        continue                                           # a pause between measurement loops
    cq = [q for s_, e_, q in mark if k1[i - 1][0] < s_ <= k1[i][0]]
    key = (k1[i][2], k2[i][2], ins[i][2], cq[-1] if cq else None)
    agg[key].append(((k1[i][0] - k1[i - 1][1]) / 1e3, (k1[i][0] - k2[i - 1][1]) / 1e3, (k1[i][1] - k1[i][0]) / 1e3))
print("queues (pre_mix, gather, insert, caller): calls | next pre_mix start after the previous pre_mix END (us, median) | after the previous gather END | pre_mix kernel us")
for key, v in agg.items():
    v.sort()
    m = v[len(v) // 2]
    a = sorted(x[1] for x in v)[len(v) // 2]
    d = sorted(x[2] for x in v)[len(v) // 2]
    print(f"  {key}: {len(v):4d} | {m[0]:8.1f} | {a:8.1f} | {d:8.1f}")
