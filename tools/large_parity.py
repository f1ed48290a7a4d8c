"""One-off large parity run: GPU engine vs oracle (multi-threaded) on many loci of a BASELINE config.
usage: python tools/large_parity.py config3 100000"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from oracle import oracle
from parity import compare, describe
from varlociraptor_amd import engine, synth
from varlociraptor_amd.batch import CallResults

name, n = sys.argv[1], int(sys.argv[2])
cfg = synth.CONFIGS[name]()
b = synth.generate(cfg, n)
plan = engine.Plan(cfg.scenario)
t0 = time.time(); got = plan.call_host(b); t_gpu = time.time() - t0
plan.close()
threads = os.cpu_count() or 8
bounds = np.linspace(0, n, threads * 4 + 1).astype(int)
oracle.lib()
t0 = time.time()
with ThreadPoolExecutor(max_workers=threads) as ex:
    parts = list(ex.map(lambda i: oracle.call(cfg.scenario, b, begin=int(bounds[i]), end=int(bounds[i + 1]), want_events=True), range(len(bounds) - 1)))
t_cpu = time.time() - t0
ref = CallResults(n, cfg.scenario.n_out, b.n_samples)
ref.event_ln_posterior = np.full((n, 1 + 2 * len(cfg.scenario.event_names)), np.nan)
for i, p in enumerate(parts):
    lo, hi = int(bounds[i]), int(bounds[i + 1])
    for f in ("ln_posterior", "map_vaf", "map_bias", "best_event", "status", "ln_marginal"):
        getattr(ref, f)[lo:hi] = getattr(p, f)[lo:hi]
    ref.event_ln_posterior[lo:hi] = p.event_ln_posterior
m = compare(got, ref, label="%s x%d" % (name, n))
print(describe(m))
print(json.dumps({"config": name, "n_loci": n, "max_abs_dposterior": m["max_dpost"], "max_abs_dmap_vaf": m["max_dvaf"], "frac_within_1e-6": m["frac_within"],
                  "bias_equal": m["bias_equal"], "best_event_equal_frac": m["best_equal_frac"], "exact_event_ties": m["n_ties"], "status_equal": m["status_equal"],
                  "oracle_seconds": round(t_cpu, 1), "oracle_threads": threads, "gpu_host_call_seconds": round(t_gpu, 2)}))
