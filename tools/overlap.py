"""tools/overlap.py -- how much the kernels of a rocprofv3 kernel trace overlap: union of busy time against the sum of durations,
and a sample of the timeline.   python tools/overlap.py trace.db [pattern]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_dc"
syms = {r[0]: r[1] for r in con.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
rows = con.execute(f"select kernel_id, start, end{', ' + qcol if qcol else ''} from rocpd_kernel_dispatch order by start").fetchall()
rows = [r for r in rows if pat in syms.get(r[0], "")]
rows = rows[len(rows) // 2:]                      # steady state: second half
tot = sum(r[2] - r[1] for r in rows)
union, cur_s, cur_e = 0, None, None
for r in rows:
    s, e = r[1], r[2]
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][2] - rows[0][1]
print(f"{len(rows)} kernels: sum of durations {tot / 1e3:.0f} us, union {union / 1e3:.0f} us, span {span / 1e3:.0f} us, overlap factor {tot / union:.2f}, idle {100 * (1 - union / span):.1f} %")
t0 = rows[0][1]
for r in rows[:18]:
    nm = syms[r[0]]
    short = "index" if "index" in nm else ("K1" if "premix" in nm else "K2")
    print(f"  {short:5s} q={r[3] if qcol else '-'} start {(r[1] - t0) / 1e3:8.1f} end {(r[2] - t0) / 1e3:8.1f} dur {(r[2] - r[1]) / 1e3:6.1f}")
