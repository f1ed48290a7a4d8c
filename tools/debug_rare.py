"""Worst |dposterior| GPU vs oracle on low-depth two-sample nested-range scenarios (found by fuzz seed 164/31)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Scenario, Sample, Contamination
from concurrent.futures import ThreadPoolExecutor
def run(label, samples, evs, depth, n=3000):
    sc = Scenario(samples, evs)
    cfg = synth.SynthConfig(name="rare", config_id=51, scenario=sc, depth=depth, type_mix={abi.VT_SNV: 0.7, abi.VT_INDEL: 0.3},
                            classes=[("c", 0.5, ((0.0, 0.2), (1.0, 1.0))), ("d", 0.5, ((0.0, 1.0), (0.5, 1.0)))], purity=None)
    b = synth.generate(cfg, n, seed=7)
    plan = engine.Plan(sc); g = plan.call_host(b); plan.close()
    th = 64; bounds = np.linspace(0, n, th + 1).astype(int)
    oracle.lib()
    with ThreadPoolExecutor(th) as ex:
        parts = list(ex.map(lambda i: oracle.call(sc, b, begin=int(bounds[i]), end=int(bounds[i + 1])), range(th)))
    ref = np.concatenate([p.ln_posterior[bounds[i]:bounds[i + 1]] for i, p in enumerate(parts)])
    d = np.nan_to_num(np.abs(np.exp(g.ln_posterior) - np.exp(ref)), nan=0.0).max(axis=1)
    q = np.sort(d)
    print("%-46s depth %4.0f: max %.2e  p99 %.2e  median %.2e  argmax depths %s" % (label, depth, d.max(), q[int(0.99 * n)], q[n // 2], tuple(b.depth()[int(d.argmax())])), flush=True)
S = lambda r, c=None: Sample(resolution=r, universe="[0.0,1.0]", contamination=c)
ev = {"y": "b:[0.0,1.0]"}
run("no contamination res .05/.02", {"a": S(0.05), "b": S(0.02)}, ev, 5.0)
run("contamination .5 res .05/.02", {"a": S(0.05, Contamination("b", 0.5)), "b": S(0.02)}, ev, 5.0)
run("contamination .1 res .05/.02", {"a": S(0.05, Contamination("b", 0.1)), "b": S(0.02)}, ev, 5.0)
run("contamination .5 res .01/.01", {"a": S(0.01, Contamination("b", 0.5)), "b": S(0.01)}, ev, 5.0)
run("contamination .5 res .05/.02", {"a": S(0.05, Contamination("b", 0.5)), "b": S(0.02)}, ev, 30.0)
run("contamination .5, b discrete", {"a": S(0.05, Contamination("b", 0.5)), "b": S(0.02)}, {"y": "b:{0.0,0.5,1.0}"}, 5.0)
run("single sample res .02", {"b": S(0.02)}, ev, 5.0)
