#!/usr/bin/env python
"""tools/rocpd_dispatches.py -- the last N kernel dispatches of a rocprofv3 rocpd database as a timeline: start / end (us from the first
one shown), queue, kernel.  Usage: rocpd_dispatches.py trace.db [N] [substring]"""
import sqlite3
import sys


def main(path, n=60, sub=""):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    syms = {r[0]: r[1] for r in cur.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
    qcol = "queue_id" if "queue_id" in cols else None
    rows = cur.execute(f"select kernel_id, start, end{', ' + qcol if qcol else ''} from rocpd_kernel_dispatch order by start").fetchall()
    rows = [r for r in rows if sub in syms.get(r[0], "")][-n:]
    t0 = rows[0][1]
    for r in rows:
        name = syms.get(r[0], str(r[0])).split("(")[0]
        print(f"{(r[1] - t0) / 1e3:10.1f} .. {(r[2] - t0) / 1e3:10.1f}  q{r[3] if qcol else '?'}  {name[-70:]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60, sys.argv[3] if len(sys.argv) > 3 else "")
