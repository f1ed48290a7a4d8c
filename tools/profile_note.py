"""Write profiles/<tag>.md (+ the bench lines and the traffic file) from what tools/profile_round.sh left in gpurun_out/<tag>/.
   python tools/profile_note.py <tag> ["free-text title"]"""
import glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else tag
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
def line(name):
    path = os.path.join(O, name)
    if not os.path.exists(path):
        return None
    rows = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(rows[-1]) if rows else None
def rocpd(d):
    db = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(O, d)) for f in fs if f.endswith(".db")]
    if not db:
        return "(no trace)\n"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db[0]], capture_output=True, text=True).stdout
    return "\n".join(out.splitlines()[:7]) + "\n"
bid = open(os.path.join(O, "build_id.txt")).read().split()[-1]
md = ["# %s" % title, "",
      "Build id `%s` (`vlr_build_id()`).  One gpurun call on a fresh MI355X box, script `tools/profile_round.sh %s`; every rocprofv3" % (bid, tag),
      "pass is its own run (`--kernel-trace --stats` for durations; `--pmc` passes with `--kernel-trace` only).  Tables made by",
      "`tools/profile_note.py` from the rocpd databases and the bench lines (committed next to this note as `%s_bench_*.json`)." % tag, "",
      "## bench.py lines (HBM-resident inputs, 3 timed launches after 1 warm-up)", "",
      "| workload | units per launch | value | ms per step | kernel ms (HIP events) | evaluations / unit | parity vs oracle | cpu_baseline port / tuned (threads = effective CPUs) |", "|---|---|---|---|---|---|---|---|"]
for w in ("config3", "config2", "config4", "config5", "realign"):
    j = line("bench_%s.json" % w)
    if not j:
        continue
    shutil.copy(os.path.join(O, "bench_%s.json" % w), os.path.join(P, "%s_bench_%s.json" % (tag, w)))
    rf = j["roofline"]; v = rf.get("valu", {})
    units = int(round(j["value"] * j["ms_per_step"] / 1e3))
    ev = v.get("pileup_evals_per_launch")
    par = j.get("parity") or {}
    pr = ("max |dposterior| %.1e, max |dMAP VAF| %.1e over %d loci" % (par.get("max_abs_dposterior", 0), par.get("max_abs_dmap_vaf", 0), par.get("n_checked", 0))) if "max_abs_dposterior" in par else ("max |dln p| %.1e over %d pairs" % (par.get("max_abs_dlnprob", 0), par.get("n_checked", 0)))
    cb = j.get("cpu_baseline") or {}
    md.append("| %s | %d | %.3f M %s | %.2f | %.2f | %s | %s | %.0f %s |" % (w, units, j["value"] / 1e6, j["unit"], j["ms_per_step"], rf["kernel_ms"],
              ("%.0f" % (ev / units)) if ev else ("%.0f cells" % (v.get("cells_per_s", 0) * rf["kernel_ms"] / 1e3 / units)), pr, cb.get("value", 0), (("/ %.0f " % cb["tuned"]["value"]) if cb.get("tuned") else "") + cb.get("unit", "") + (" (%d threads)" % cb.get("cores", 0))))
j = line("bench_config3_afd.json")
if j and j.get("with_afd"):
    shutil.copy(os.path.join(O, "bench_config3_afd.json"), os.path.join(P, "%s_bench_config3_afd.json" % tag))
    a = j["with_afd"]
    md += ["", "With AFD lists (`bench.py --afd`, capacity %d): %.3f M loci/s = %.0f %% of the plain rate (%.1f ms for call pass + log filter + replay of overflowed loci), mean %.1f points per sample list, %d truncated lists."
           % (a["afd_capacity"], a["value"] / 1e6, 100 * a["ratio_to_plain"], a["kernel_ms_both_launches"], a["mean_afd_points_per_sample"], a["truncated_lists"])]
j = line("bench_cli.json")
if j:
    shutil.copy(os.path.join(O, "bench_cli.json"), os.path.join(P, "%s_bench_cli.json" % tag))
    st, nt = j["stages_s"], (j.get("native_stage_seconds_per_step") or j.get("native_stage_seconds_last_step") or {})
    md += ["", "End to end through the process boundary (`bench.py --workload cli`: %s; %d effective CPUs of %d visible): **%.1f k records/s** — read %.2f s (inflate %.2f summed over the two files, both files in %.2f wall, merge %.2f), call %.2f s, write %.2f s (encode %.2f, deflate + file %.2f)."
           % (j["config"]["workload"], j["config"].get("effective_cpus", 0), j["config"].get("host_threads", 0), j["value"] / 1e3, st["read_s"], nt.get("inflate", 0), nt.get("files_wall", 0), nt.get("merge", 0), st["call_s"], st["write_s"], nt.get("encode", 0), nt.get("deflate_write", 0))]
for fn, what in (("bench_cli_1M.json", "the same with 1 000 000 records per step (the size of BASELINE configs[2])"), ("bench_cli_hostreader.json", "the same through the HOST reader (`VLR_INGEST_HOST=1`: libdeflate + v15 decode on the CPUs)")):
    j = line(fn)
    if j:
        shutil.copy(os.path.join(O, fn), os.path.join(P, "%s_%s" % (tag, fn)))
        st = j["stages_s"]
        md += ["", "… %s: **%.1f k records/s** — read %.2f s, call %.2f s, write %.2f s per step." % (what, j["value"] / 1e3, st["read_s"], st["call_s"], st["write_s"])]
j = line("bench_ingest.json")
if j:
    shutil.copy(os.path.join(O, "bench_ingest.json"), os.path.join(P, "%s_bench_ingest.json" % tag))
    st, rf, fl = j["stages_s"], j["roofline"], j["files"]
    md += ["", "Device reader alone (`bench.py --workload ingest`: %s): **%.1f k records/s** (host reader on the same files: %s records/s); per step: members up + inflate wait %.3f s, record split + INFO scan %.3f, decode %.3f, column copy %.3f, host side %.3f, all %.3f; %.2f GB inflated from %.2f GB; inflate kernels %.1f ms per step = **%.1f GB/s** of algorithmic bytes (compressed in + inflated out) = %.2f %% of the HBM roofline; %d chunks fell back to the serial record walk."
           % (j["config"]["workload"], j["value"] / 1e3, ("%.0f k" % (j["cpu_baseline"]["value"] / 1e3)) if j.get("cpu_baseline") else "n/a", st.get("feed_inflate", 0), st.get("split_scan", 0), st.get("decode", 0),
              st.get("copy_back", 0), st.get("host_table", 0), st.get("total", 0), fl["inflated_bytes"] / 1e9, fl["observation_bcf_bytes"] / 1e9, rf["kernel_ms"], rf["achieved"] or 0, 100 * (rf["frac"] or 0), fl["serial_walks"])]
pi = os.path.join(O, "pmc_inflate.md")
if os.path.exists(pi):
    md += ["", "## Kernels of the device reader: rocprofv3 --kernel-trace --stats and PMC passes of `tools/ingest_rate.py 50000 32768 device` (3 passes over 2 x 50 000 records; `tools/pmc_inflate.sh`; counters of vlr_inflate_kernel, summed over dispatches)", "", open(pi).read().rstrip(), ""]
if os.path.exists(os.path.join(O, "cpu_probe.txt")):
    shutil.copy(os.path.join(O, "cpu_probe.txt"), os.path.join(P, "%s_cpu_probe.txt" % tag))
md += ["", "## rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline` (config3, 1 M loci per launch)", "", rocpd("stats_config3"),
       "## … of `python bench.py --afd --no-cpu-baseline`", "", rocpd("stats_config3_afd"),
       "## … of `python bench.py --workload realign --no-cpu-baseline`", "", rocpd("stats_realign")]
sc = os.path.join(O, "stats_cli.md")
if os.path.exists(sc):
    md += ["## … of `python bench.py --workload cli --steps 3 --warmup 1` (the whole pipeline: reader, evaluation, emission kernels on their streams; durations of kernels on the low-priority feed stream include the time they wait for CUs) and the union of the kernel intervals over the timed steps (`tools/timeline_busy.py`)", "", open(sc).read().rstrip(), ""]
md += ["## PMC counters per locus (= per wave; config3, 50 000 loci, `tools/pmc_pass.sh`)", ""]
for name in ("insts", "util", "f64", "icache"):
    f = os.path.join(O, "pmc", name + ".md")
    if os.path.exists(f):
        md += [open(f).read().rstrip(), ""]
tj = os.path.join(ROOT, "gpurun_out", "traffic_config3.json")
if os.path.exists(tj):
    t = json.load(open(tj))
    if t.get("build_id") == bid:
        shutil.copy(tj, os.path.join(P, "traffic_config3.json"))
        md += ["## HBM traffic per 1 M-locus launch (`tools/traffic_measure.sh config3` -> profiles/traffic_config3.json)", "",
               "    FETCH_SIZE %.0f KB raw x calibration %.3f (4 B/lane stream of known size) = %.2f GB read" % (t["FETCH_SIZE_KB_per_launch"], t["calibration"]["read_4B_per_lane"]["factor"], t["read_bytes_per_launch"] / 1e9),
               "    WRITE_SIZE %.0f KB raw x calibration %.3f (8 B/lane stream of known size) = %.2f GB written" % (t["WRITE_SIZE_KB_per_launch"], t["calibration"]["write_8B_per_lane"]["factor"], t["write_bytes_per_launch"] / 1e9),
               "    total %.2f GB vs %.2f GB algorithmic = %.2f x" % (t["hbm_bytes_per_launch"] / 1e9, t["algorithmic_bytes_per_launch"] / 1e9, t["ratio_to_algorithmic"]), ""]
extra = os.path.join(O, "notes.md")
if os.path.exists(extra):
    md += [open(extra).read()]
open(os.path.join(P, tag + ".md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
