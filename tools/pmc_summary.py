#!/usr/bin/env python
"""Per-kernel PMC counter sums of a rocprofv3 --pmc run (ROCm 7.2 rocpd sqlite output), as a markdown table.

usage: python tools/pmc_summary.py <results.db> [divide_by] [kernel name pattern]     (divide_by: e.g. the number of loci = waves)
"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    like = sys.argv[3] if len(sys.argv) > 3 else "vlr_call_kernel"   # kernel name pattern
    rows = list(con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                            "where kernel_name like ? and kernel_name not like '%vlr_deep%' group by kernel_name, counter_name order by counter_name", ("%" + like + "%",)))
    print("| counter | sum over dispatches | dispatches | per unit (/%g) |" % div)
    print("|---|---|---|---|")
    for _, name, val, nd in rows:
        print("| %s | %.0f | %d | %.1f |" % (name, val, nd, val / max(nd, 1) / div))


if __name__ == "__main__":
    main()
