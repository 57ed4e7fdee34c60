#!/usr/bin/env python
"""tools/rocpd_stats.py -- per-kernel summary (calls, avg/min/max us, % of GPU time) from a
rocprofv3 rocpd SQLite database (ROCm 7.2 default output).  Usage: rocpd_stats.py trace.db [out.csv]"""
import sqlite3
import sys


def main(path, out=None):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    rows = cur.execute("select kernel_id, start, end from rocpd_kernel_dispatch").fetchall()
    agg = {}
    for kid, st, en in rows:
        name = syms.get(kid, str(kid))
        a = agg.setdefault(name, [])
        a.append((en - st) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    lines = ["name,calls,total_us,avg_us,min_us,max_us,pct"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0][:90]
        lines.append(f"\"{short}\",{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.1f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
