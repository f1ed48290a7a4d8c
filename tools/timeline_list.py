#!/usr/bin/env python
"""The kernels of a rocprofv3 trace (rocpd sqlite) in start order: start and end in ms from the first one, duration, queue and stream
ids, name.   usage: python tools/timeline_list.py <results.db> [first [count]]"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "grid_x") if c in cols]
rows = list(cur.execute("select %s from kernels order by start" % ", ".join(want)))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
t0 = rows[0][want.index("start")]
print("columns of the kernels view:", cols)
for r in rows[first:first + count]:
    d = dict(zip(want, r))
    print("%9.3f %9.3f %7.3f ms  q%-3s s%-3s %8s  %s" % ((d["start"] - t0) / 1e6, (d["end"] - t0) / 1e6, (d["end"] - d["start"]) / 1e6, d.get("queue_id", "?"), d.get("stream_id", "?"), d.get("grid_x", "?"),
                                                    d["name"].split("(")[0][-44:]))
