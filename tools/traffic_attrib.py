"""Where the HBM read traffic of the call kernel goes: FETCH_SIZE of single launches on variants of config 3 —
   A standard, B without artifact hypotheses (bias mask 0: one hypothesis per locus), C SNVs only (no third coefficient e, no
   scratch row), D both — and the algorithmic column bytes of each.  Run under
       rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o p -- python tools/traffic_attrib.py run
   then  python tools/traffic_attrib.py report <dir>  (reads factor 2.000 for 4 B/lane streams, tools/traffic_calibrate.py)."""
import glob, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 200000
VARIANTS = [("A standard", None, None), ("B one hypothesis", 0, None), ("C SNV only", None, "snv"), ("D one hypothesis, SNV only", 0, "snv")]


def run():
    import torch
    from varlociraptor_amd import abi, engine, synth
    out = []
    for name, mask, mix in VARIANTS:
        cfg = synth.config3(type_mix={abi.VT_SNV: 1.0} if mix == "snv" else None)
        b = synth.generate(cfg, N, seed=77, bias_mask=abi.BIAS_ALL if mask is None else mask)
        plan = engine.Plan(cfg.scenario)
        plan.set_max_obs(int(b.depth().sum(axis=1).max()))
        db = engine.DeviceBatch(b, "cuda:0")
        res = engine.DeviceResults(b.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        plan.call_device(db, res, st)
        torch.cuda.synchronize()
        out.append({"variant": name, "n_loci": N, "n_obs": int(b.n_obs), "column_bytes": int(b.n_obs) * 40 + N * 2 * 4, "kernel_ms": plan.last_kernel_ms(),
                    "evals": plan.work_counters()[0]})
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic_attrib_run.json"), "w"), indent=1)


def report(d):
    runs = json.load(open(os.path.join(ROOT, "gpurun_out", "traffic_attrib_run.json")))
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    rows = list(con.execute("select dispatch_id, sum(value) from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like '%vlr_call_kernel%' "
                            "and kernel_name not like '%vlr_deep%' group by dispatch_id order by dispatch_id"))
    assert len(rows) == len(runs), (len(rows), len(runs))
    for r, (_, kb) in zip(runs, rows):
        r["read_bytes"] = kb * 1024.0 * 2.0
        r["read_over_columns"] = r["read_bytes"] / r["column_bytes"]
        print("%-28s obs %9d  columns %.3f GB  read %.3f GB  = %.2f x   %.1f ms  %d evals" % (r["variant"], r["n_obs"], r["column_bytes"] / 1e9, r["read_bytes"] / 1e9, r["read_over_columns"], r["kernel_ms"], r["evals"]))
    json.dump(runs, open(os.path.join(ROOT, "gpurun_out", "traffic_attrib.json"), "w"), indent=1)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
