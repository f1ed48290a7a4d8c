import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from varlociraptor_amd import synth, engine
from oracle import oracle
from parity import compare, describe
from test_gpu_edge_cases import oracle_mt
from bench import generate
cfg = synth.config2()
b = generate("config2", 100000, 0, chunk_loci=25000, workers=8)
plan = engine.Plan(cfg.scenario)
got = plan.call_host(b)
ref = oracle_mt(oracle, cfg.scenario, b, threads=64)
m = compare(got, ref, label="c2")
print(describe(m))
bad = np.nonzero((got.map_bias != ref.map_bias).any(1) | (got.status != ref.status))[0]
print(len(bad), bad[:10])
for l in bad[:4]:
    print(" locus", l, "depth", b.depth()[l], "flags", bin(b.locus["locus_flags"][l]))
    print("  got post", np.exp(got.ln_posterior[l]), "map", got.map_vaf[l], "bias", got.map_bias[l], "best", got.best_event[l], "st", got.status[l])
    print("  ref post", np.exp(ref.ln_posterior[l]), "map", ref.map_vaf[l], "bias", ref.map_bias[l], "best", ref.best_event[l], "st", ref.status[l])
    print("  ref events", ref.event_ln_posterior[l])
    sl = b.pileup_slice(l, 0)
    print("  pa", b.columns["prob_alt"][sl][:40]); print("  pr", b.columns["prob_ref"][sl][:40]); print("  fl", [hex(x) for x in b.columns["flags"][sl][:40]])
