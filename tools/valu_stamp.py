"""Turn the PMC passes of tools/pmc_pass.sh (insts, util, f64) into profiles/valu_<workload>.json: instructions per locus of the
call kernel, stamped with the build id, for the `valu_busy` / `f64_share` keys of bench.py's roofline object (VERDICT r05 "next" #4).

usage: python tools/valu_stamp.py <dir of the pmc passes> <workload> <loci of the passes>     -> gpurun_out/valu_<workload>.json
"""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, W, N = sys.argv[1], sys.argv[2], int(sys.argv[3])


def counters(name):
    db = glob.glob(os.path.join(O, name, "**", "*.db"), recursive=True)
    if not db:   # pmc_pass.sh deletes large databases after the summary: read the markdown table it leaves
        out = {}
        for l in open(os.path.join(O, name + ".md")):
            f = [x.strip() for x in l.strip().strip("|").split("|")]
            if len(f) == 4 and f[0] not in ("counter", "---"):
                out[f[0]] = float(f[3])
        return out
    con = sqlite3.connect(db[0])
    rows = con.execute("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%vlr_call_kernel%' "
                       "and kernel_name not like '%vlr_deep%' group by counter_name")
    return {n: v / max(d, 1) / N for n, v, d in rows}


ins, util, f64 = counters("insts"), counters("util"), counters("f64")
line = json.loads(open(os.path.join(O, "insts.json")).read().strip().splitlines()[-1])
valu = ins["SQ_INSTS_VALU"]
f64_arith = f64["SQ_INSTS_VALU_FMA_F64"] + f64["SQ_INSTS_VALU_MUL_F64"] + f64["SQ_INSTS_VALU_ADD_F64"]
out = {
    "workload": W, "n_units": N, "build_id": line["build_id"],
    "command": "tools/pmc_pass.sh <dir> %s %d (rocprofv3 --pmc, one pass per counter group, --kernel-trace only)" % (W, N),
    "per_locus": {"valu_insts": valu, "salu_insts": ins["SQ_INSTS_SALU"], "lds_insts": ins["SQ_INSTS_LDS"], "smem_insts": ins["SQ_INSTS_SMEM"],
                  "f64_fma": f64["SQ_INSTS_VALU_FMA_F64"], "f64_mul": f64["SQ_INSTS_VALU_MUL_F64"], "f64_add": f64["SQ_INSTS_VALU_ADD_F64"],
                  "f64_trans": f64.get("SQ_INSTS_VALU_TRANS_F64", 0.0), "wave_cycles": ins.get("SQ_WAVE_CYCLES")},
    "f64_share": f64_arith / valu,
    "lane_utilisation": (util["SQ_THREAD_CYCLES_VALU"] / (64.0 * util["SQ_ACTIVE_INST_VALU"])) if util.get("SQ_ACTIVE_INST_VALU") else None,
    "note": "one wave per locus: per-locus = per-wave instruction counts; the figures are averages over the workload's loci (same generator and seed as the bench line)",
}
path = os.path.join(ROOT, "gpurun_out", "valu_%s.json" % W)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
