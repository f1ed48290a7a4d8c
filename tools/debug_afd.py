import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import synth, engine
from oracle import oracle
cfg = synth.CONFIGS[sys.argv[1]](); n = int(sys.argv[2]); seed = int(sys.argv[3])
b = synth.generate(cfg, n, seed=seed)
plan = engine.Plan(cfg.scenario)
got = plan.call_host(b, afd_capacity=256)
ref = oracle.call(cfg.scenario, b, afd_capacity=256, want_events=True)
bad = 0
for l in range(n):
    for s in range(b.n_samples):
        if got.afd_count[l, s] != ref.afd_count[l, s]:
            bad += 1
            if bad > 3: continue
            print("locus", l, "sample", s, "map", got.map_vaf[l], ref.map_vaf[l], "best", got.best_event[l], ref.best_event[l], "depth", b.depth()[l])
            for nm, r in (("gpu", got), ("ref", ref)):
                k = r.afd_count[l, s]
                o = np.lexsort((r.afd_lnprob[l, s, :k], r.afd_vaf[l, s, :k]))
                print(" ", nm, [(float(r.afd_vaf[l, s, i]), round(float(r.afd_lnprob[l, s, i]), 4)) for i in o])
print("mismatching lists:", bad)
