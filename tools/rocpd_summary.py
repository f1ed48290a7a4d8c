#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as the `--stats` table.

usage: python tools/rocpd_summary.py <results.db> [out.md]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % | vgpr | sgpr | lds B | scratch B | grid | wg |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        lines.append("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
