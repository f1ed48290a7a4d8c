"""Re-create fuzz scenario (seed, it), select loci, and compare GPU/oracle under single-bias masks."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Scenario
src = open("tools/fuzz_scenarios.py").read()
fz = type(sys)("fz"); fz.__file__ = os.path.abspath("tools/fuzz_scenarios.py"); exec(compile(src.split("def main()")[0], "fz", "exec"), fz.__dict__)
seed, target, locus = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
events = eval(sys.argv[4]) if len(sys.argv) > 4 else None
rng = np.random.default_rng(seed)
for it in range(target + 1):
    try:
        sc, names = fz.random_scenario(rng); sc.desc()
    except Exception:
        continue
    S=len(names)
    classes=[]
    for _ in range(4):
        classes.append(("c", 0.25, tuple((float(v), float(v + w)) for v, w in zip(rng.choice([0.0, 0.1, 0.5, 1.0], S), rng.choice([0.0, 0.0, 0.2], S)))))
    classes = [(l, f, tuple((lo, min(hi, 1.0)) for lo, hi in spec)) for l, f, spec in classes]
    cfg = synth.SynthConfig(name="fuzz", config_id=50, scenario=sc, depth=float(rng.choice([4.0, 12.0, 30.0])), type_mix={abi.VT_SNV: 0.7, abi.VT_INDEL: 0.3}, classes=classes, purity=None)
    b = synth.generate(cfg, 24, seed=int(rng.integers(1 << 30)))
if events: sc = Scenario(sc.samples, events)
sub = b.select([locus])
np.set_printoptions(precision=6, linewidth=220)
print(sc.events, "variant type", sub.locus["variant_type"], "flags", bin(int(sub.locus["locus_flags"][0])))
if os.environ.get("AFD"):
    plan=engine.Plan(sc); g=plan.call_host(sub, afd_capacity=128); plan.close()
    r=oracle.call(sc,sub,afd_capacity=128)
    for si in range(sub.n_samples):
        ng, nr = int(g.afd_count[0, si]), int(r.afd_count[0, si])
        print("AFD s%d gpu" % si, ng, list(zip(g.afd_vaf[0, si, :min(ng,128)].round(5), g.afd_lnprob[0, si, :min(ng,128)].round(6))))
        print("AFD s%d ref" % si, nr, list(zip(r.afd_vaf[0, si, :min(nr,128)].round(5), r.afd_lnprob[0, si, :min(nr,128)].round(6))))
for mask in [abi.BIAS_ALL, abi.BIAS_STRAND, abi.BIAS_ORIENTATION, abi.BIAS_POSITION, abi.BIAS_SOFTCLIP, abi.BIAS_HOMOPOLYMER, abi.BIAS_ALTLOCUS]:
    sub.locus["locus_flags"][:] = (sub.locus["locus_flags"] & ~np.uint8(0x3f)) | np.uint8(mask)
    plan=engine.Plan(sc); g=plan.call_host(sub); plan.close()
    r=oracle.call(sc,sub,want_events=True)
    print("mask %02x gpu" % mask, g.ln_posterior[0], "| ref", r.ln_posterior[0], "| ref events", r.event_ln_posterior[0])
    print("   map gpu", g.map_vaf[0], "best", g.best_event[0], "bias", g.map_bias[0], "| ref", r.map_vaf[0], "best", r.best_event[0], "bias", r.map_bias[0])
if os.environ.get("DUMP"):
    for s in range(sub.n_samples):
        sl = sub.pileup_slice(0, s)
        f = sub.columns["flags"][sl]
        strand = (f >> abi.F_STRAND_SHIFT) & 3
        pa, pr, pm = sub.columns["prob_alt"][sl], sub.columns["prob_ref"][sl], sub.columns["prob_mapping"][sl]
        bf_ref = np.exp(pr.astype(np.float64) - pa); bf_alt = np.exp(pa.astype(np.float64) - pr)
        print("sample", s, "n", len(pa))
        print("  strand", strand, "\n  strong_ref", (bf_ref > 20).astype(int), "\n  strong_alt", (bf_alt > 20).astype(int), "\n  pm", pm, "\n  uniq", (pm >= np.log(0.95)).astype(int))
        print("  orient", (f >> abi.F_ORIENT_SHIFT) & 3)
