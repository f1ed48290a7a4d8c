#!/usr/bin/env python
"""GPU busy time of a rocprofv3 kernel trace (rocpd sqlite): the union of all kernel intervals, per kernel name the time it runs alone
or beside others, and the idle gaps — says whether a pipeline of several streams is bound by the device.

usage: python tools/timeline_busy.py <results.db> [t0_fraction t1_fraction]   (fractions of the traced span to look at, default 0 1)
"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if "start" not in cols or "end" not in cols:
        print("kernels view has columns", cols)
        return
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        print("no kernels")
        return
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
    t_min, t_max = rows[0][1], max(r[2] for r in rows)
    lo, hi = t_min + (t_max - t_min) * f0, t_min + (t_max - t_min) * f1
    rows = [(n, max(s, lo), min(e, hi)) for n, s, e in rows if e > lo and s < hi]
    # sweep: events
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    last = lo
    busy = 0
    alone = {}
    shared = {}
    depth_time = {}
    for t, kind, i in ev:
        dt = t - last
        if dt > 0 and active:
            busy += dt
            depth_time[len(active)] = depth_time.get(len(active), 0) + dt
            for j in active:
                name = rows[j][0].split("(")[0][-40:]
                if len(active) == 1:
                    alone[name] = alone.get(name, 0) + dt
                else:
                    shared[name] = shared.get(name, 0) + dt
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    span = hi - lo
    print("span %.1f ms, some kernel running %.1f ms (%.0f %%), idle %.1f ms" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    print("kernels in flight: " + ", ".join("%d: %.1f ms" % (k, v / 1e6) for k, v in sorted(depth_time.items())))
    print("| kernel | alone ms | beside others ms |")
    print("|---|---|---|")
    for name in sorted(set(alone) | set(shared), key=lambda n: -(alone.get(n, 0) + shared.get(n, 0))):
        print("| %s | %.1f | %.1f |" % (name, alone.get(name, 0) / 1e6, shared.get(name, 0) / 1e6))


if __name__ == "__main__":
    main()
