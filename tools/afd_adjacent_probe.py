"""AFD lists of scenarios made of ADJACENT ranges of one sample (shared end points are visited from both neighbours): engine vs oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from varlociraptor_amd import engine, synth
from varlociraptor_amd.scenario import Scenario, Sample
from oracle import oracle
for n_ev in (5, 10, 30, 45):
    edges = [round(k / float(n_ev), 6) for k in range(n_ev + 1)]
    sc = Scenario({"s": Sample(resolution=0.01, universe="[0.0,1.0]")}, {"e%02d" % k: "s:]%s,%s]" % (repr(edges[k]), repr(edges[k + 1])) for k in range(n_ev)})
    cfg = synth.config2(); cfg.depth = 30.0; cfg.scenario = sc
    b = synth.generate(cfg, 60, seed=29)
    plan = engine.Plan(sc); g = plan.call_host(b, afd_capacity=1024); plan.close()
    r = oracle.call(sc, b, afd_capacity=1024, want_events=True)
    bad = [(l, int(g.afd_count[l, 0]), int(r.afd_count[l, 0])) for l in range(b.n_loci) if g.best_event[l] == r.best_event[l] and g.afd_count[l, 0] != r.afd_count[l, 0]]
    print(n_ev, "events: loci with different AFD counts:", len(bad), bad[:5])
    if bad:
        l = bad[0][0]
        gv = np.sort(np.asarray(g.afd_vaf[l, 0][:g.afd_count[l, 0]])); rv = np.sort(np.asarray(r.afd_vaf[l, 0][:r.afd_count[l, 0]]))
        u, cnt = np.unique(gv, return_counts=True)
        print("   engine duplicates:", u[cnt > 1][:12], "engine-only:", np.setdiff1d(gv, rv)[:8], "oracle-only:", np.setdiff1d(rv, gv)[:8])
