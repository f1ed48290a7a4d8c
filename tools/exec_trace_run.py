"""Drivers of the round-6 diagnosis builds of the call kernel (csrc/vlr_kernels.hip: VLR_DBG_EXEC_ASSERT, VLR_DBG_TRACE; built by
tools/dbg_variant.sh, run on the GPU box by tools/exec_assert.sh).  The library comes from VLR_LIB like everywhere else.

  python tools/exec_trace_run.py assert out.json [quick|full]      run the build-matrix workloads, dump the per-site EXEC / uniformity table
  python tools/exec_trace_run.py first a.npz b.npz                 first (workload, locus) whose results differ between two matrix dumps
  python tools/exec_trace_run.py trace <workload> <locus> out.npz  trace the wave of one locus of one matrix workload
  python tools/exec_trace_run.py diff a.npz b.npz [n]              first n records at which two traces part
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

N_SITES = 8192


def run_workloads(mode, only=None):
    import matrix_run
    from varlociraptor_amd import engine
    out = {}
    for name, sc, batch, afd in matrix_run.workloads(mode):
        if only is not None and name != only:
            continue
        try:
            plan = engine.Plan(sc)
        except Exception:
            continue
        out[name] = plan.call_host(batch, afd_capacity=afd)
        plan.close()
    return out


def cmd_assert(outp, mode):
    from varlociraptor_amd import engine
    L = engine.lib()
    L.vlr_debug_exec_sites.restype = C.c_int
    L.vlr_debug_exec_sites.argtypes = [C.c_void_p, C.c_int]
    run_workloads(mode)
    tab = np.zeros(N_SITES * 4, dtype=np.uint64)
    rc = L.vlr_debug_exec_sites(tab.ctypes.data, 0)
    assert rc == 0, rc
    tab = tab.reshape(N_SITES, 4)
    src = open(os.path.join(ROOT, "varlociraptor_amd", "csrc", "vlr_kernels.hip")).read().split("\n")
    rows = []
    for line in range(N_SITES):
        ex, part, viol, missing = (int(x) for x in tab[line])
        if ex:
            rows.append({"line": line, "partial": part, "violations": viol, "disabled_lanes": "%016x" % missing,
                         "source": src[line - 1].strip()[:160] if 0 < line <= len(src) else ""})
    res = {"lib": os.environ.get("VLR_LIB", "default"), "build_id": engine.build_id(), "mode": mode, "sites_executed": len(rows),
           "sites_partial_exec": sum(1 for r in rows if r["partial"]), "sites_with_violations": sum(1 for r in rows if r["violations"]), "sites": rows}
    json.dump(res, open(outp, "w"), indent=1)
    print("%s: %d sites executed, %d under partial EXEC at least once, %d with rule violations" %
          (res["lib"], res["sites_executed"], res["sites_partial_exec"], res["sites_with_violations"]))
    for r in rows:
        if r["violations"]:
            print("  VIOLATION line %5d  x%-8d partial x%-8d disabled %s   %s" % (r["line"], r["violations"], r["partial"], r["disabled_lanes"], r["source"]))
    for r in rows:
        if r["partial"] and not r["violations"]:
            print("  partial   line %5d  x%-8d disabled %s   %s" % (r["line"], r["partial"], r["disabled_lanes"], r["source"]))


def first_diff(a, b):
    A, B = np.load(a), np.load(b)
    best = None
    order = []
    for k in A.files:
        if k == "build_id" or k not in B.files:
            continue
        x, y = A[k], B[k]
        if x.shape != y.shape:
            continue
        xv = x.view(np.uint8) if x.dtype.kind == "f" else x
        yv = y.view(np.uint8) if y.dtype.kind == "f" else y
        if np.array_equal(xv, yv):
            continue
        rows = np.nonzero((xv.reshape(len(x), -1) != yv.reshape(len(y), -1)).any(axis=1))[0]
        order.append((k, int(rows[0]), len(rows)))
    # prefer the simplest workloads: single-sample pileups first
    pref = ["single_1.5", "single_3", "single_7", "single_12", "tn_tiny", "config2", "config3", "config5", "config4"]
    for w in pref:
        for k, r, n in order:
            if k.split("/")[0] == w and k.endswith("/ln_posterior"):
                return w, r, order
    if order:
        return order[0][0].split("/")[0], order[0][1], order
    return None, None, order


def cmd_trace(workload, locus, outp):
    from varlociraptor_amd import engine
    L = engine.lib()
    L.vlr_debug_trace_arm.restype = C.c_int
    L.vlr_debug_trace_arm.argtypes = [C.c_longlong]
    L.vlr_debug_trace_read.restype = C.c_longlong
    L.vlr_debug_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    assert L.vlr_debug_trace_arm(int(locus)) == 0
    got = run_workloads("quick", only=workload)
    cap = 1 << 16
    hdr = np.zeros(2 * cap, dtype=np.uint64)
    val = np.zeros(64 * cap, dtype=np.float64)
    n = L.vlr_debug_trace_read(hdr.ctypes.data, val.ctypes.data, cap)
    assert n >= 0, n
    k = min(n, cap)
    L.vlr_debug_trace_arm(-1)
    g = got[workload]
    np.savez(outp, hdr=hdr[:2 * k].reshape(k, 2), val=val[:64 * k].reshape(k, 64), n=np.array([n]),
             ln_posterior=np.asarray(g.ln_posterior)[locus], status=np.asarray(g.status)[locus])
    print("trace of %s locus %d: %d records (%s), ln_posterior %s" % (workload, locus, n, os.environ.get("VLR_LIB", "default"), np.asarray(g.ln_posterior)[locus]))


def cmd_diff(a, b, nshow=12):
    A, B = np.load(a), np.load(b)
    ha, hb, va, vb = A["hdr"], B["hdr"], A["val"], B["val"]
    print("records: %d vs %d; results %s vs %s" % (len(ha), len(hb), A["ln_posterior"], B["ln_posterior"]))
    src = open(os.path.join(ROOT, "varlociraptor_amd", "csrc", "vlr_kernels.hip")).read().split("\n")
    shown = 0
    n = min(len(ha), len(hb))
    for r in range(n):
        ida, la, ea = int(ha[r, 0]) >> 32, int(ha[r, 0]) & 0xffffffff, int(ha[r, 1])
        idb, lb, eb = int(hb[r, 0]) >> 32, int(hb[r, 0]) & 0xffffffff, int(hb[r, 1])
        if (ida, la) != (idb, lb):
            print("record %d: the traces take different paths here: id %d line %d  vs  id %d line %d" % (r, ida, la, idb, lb))
            for rr in range(max(0, r - 6), min(n, r + 3)):
                print("      %6d: a id %3d line %4d | b id %3d line %4d" % (rr, int(ha[rr, 0]) >> 32, int(ha[rr, 0]) & 0xffffffff, int(hb[rr, 0]) >> 32, int(hb[rr, 0]) & 0xffffffff))
            break
        act = np.array([(ea >> l) & 1 for l in range(64)], dtype=bool) & np.array([(eb >> l) & 1 for l in range(64)], dtype=bool)
        xa, xb = va[r].view(np.uint64), vb[r].view(np.uint64)
        bad = np.nonzero((xa != xb) & act)[0]
        if ea != eb or len(bad):
            shown += 1
            print("record %d id %d line %d (%s): EXEC %016x vs %016x; %d active lanes differ" % (r, ida, la, src[la - 1].strip()[:90], ea, eb, len(bad)))
            for l in bad[:8]:
                print("        lane %2d: %r (%016x)  vs  %r (%016x)" % (l, float(va[r, l]), int(xa[l]), float(vb[r, l]), int(xb[l])))
            if shown >= nshow:
                break
    else:
        if len(ha) != len(hb):
            print("one trace is a prefix of the other (%d vs %d records)" % (len(ha), len(hb)))
    if shown == 0:
        print("no differing record among the first %d" % n)


if __name__ == "__main__":
    if sys.argv[1] == "assert":
        cmd_assert(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "quick")
    elif sys.argv[1] == "first":
        w, r, order = first_diff(sys.argv[2], sys.argv[3])
        print("%d arrays differ" % len(order))
        for k, row, n in order[:10]:
            print("    %s: first row %d, %d rows" % (k, row, n))
        print("FIRST %s %s" % (w, r))
    elif sys.argv[1] == "trace":
        cmd_trace(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "diff":
        cmd_diff(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 12)
