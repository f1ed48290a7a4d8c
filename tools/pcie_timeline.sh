#!/bin/bash
# Kernel + memory-copy trace of vlr_batch_run_host on page-locked arrays (tools/pcie_rate.py): do the chunk copies run beside the kernels?
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pcie_tl; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/t -o t -- python $R/tools/pcie_rate.py ${1:-400000} > $O/out.txt 2> $O/err.txt)
tail -3 $O/out.txt
DB=$(find $O/t -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
mc = [t for t in tabs if "memory_cop" in t.lower()]
print("copy views:", mc[:4])
kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ks = list(cur.execute("select name, start, end from kernels order by start"))
v = mc[0]
cc = [r[1] for r in cur.execute("pragma table_info(%s)" % v)]
print(cc)
cs = list(cur.execute("select * from %s order by start" % v))
si, ei = cc.index("start"), cc.index("end")
szi = cc.index("size") if "size" in cc else None
ni = cc.index("name") if "name" in cc else None
t0 = min(ks[0][1], cs[0][si])
ev = [((k[1]-t0)/1e6, (k[2]-t0)/1e6, "K " + k[0].split("(")[0][-30:]) for k in ks if (k[2]-k[1]) > 5e5]
ev += [((c[si]-t0)/1e6, (c[ei]-t0)/1e6, "C %s %s MB" % (c[ni] if ni is not None else "", (c[szi] >> 20) if szi is not None else "?")) for c in cs if (c[ei]-c[si]) > 5e5]
ev.sort()
# the last host call (page-locked arrays): the final third of the events
for e in ev[-70:]: print("%9.2f %9.2f %7.2f  %s" % (e[0], e[1], e[1]-e[0], e[2]))
PY
