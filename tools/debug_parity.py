import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from varlociraptor_amd import synth, engine, abi
from oracle import oracle
from parity import compare, describe
np.set_printoptions(precision=6, linewidth=200)

def run(cfg, b, label):
    plan = engine.Plan(cfg.scenario)
    got = plan.call_host(b)
    ref = oracle.call(cfg.scenario, b, want_events=True)
    m = compare(got, ref, label=label)
    print(describe(m))
    bad = list(m["bad"]) + [int(i) for i in np.nonzero((got.map_bias != ref.map_bias).any(1))[0]]
    for l in bad[:6]:
        print(" locus", l, "depth", b.depth()[l], "flags", bin(b.locus["locus_flags"][l]), "vt", b.locus["variant_type"][l])
        print("  got post", np.exp(got.ln_posterior[l]), "map", got.map_vaf[l], "bias", got.map_bias[l], "best", got.best_event[l], "st", got.status[l])
        print("  ref post", np.exp(ref.ln_posterior[l]), "map", ref.map_vaf[l], "bias", ref.map_bias[l], "best", ref.best_event[l], "st", ref.status[l])
        print("  ref events", ref.event_ln_posterior[l])
        for s in range(b.n_samples):
            sl = b.pileup_slice(l, s)
            print("  s%d pa" % s, b.columns["prob_alt"][sl][:12], "pr", b.columns["prob_ref"][sl][:12], "pm", b.columns["prob_mapping"][sl][:2])
            print("     flags", [hex(x) for x in b.columns["flags"][sl][:12]])

for depth, seed in [(7.0, 11), (12.0, 11)]:
    cfg = synth.config2(); cfg.depth = depth
    run(cfg, synth.generate(cfg, 300, seed=seed), "single %.1f" % depth)
cfg = synth.config3(); cfg.depth = 4.0; cfg.empty_fraction = 0.25
run(cfg, synth.generate(cfg, 300, seed=12), "tn4")
