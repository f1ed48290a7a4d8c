import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Sample, Scenario, Conj, Atom, Lfc, VAFSet, Contamination, parse_formula
import importlib.util
spec = importlib.util.spec_from_file_location("fz", "tools/fuzz_scenarios.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
rng = np.random.default_rng(1)
for it in range(92):
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
    if it == 91: break
print(sc.events, {k:(v.resolution, v.contamination) for k,v in sc.samples.items()})
sub = b.select([6])
A = 0.0 + 0.1*3
def run(ev):
    s2=Scenario(sc.samples, ev)
    plan=engine.Plan(s2); g=plan.call_host(sub); plan.close()
    r=oracle.call(s2,sub,want_events=True)
    return g.ln_posterior[0,1]+g.ln_marginal[0], r.ln_posterior[0,1]+r.ln_marginal[0]
print("chain gpu/ref", run({"r": Conj([parse_formula("c:0.5"), Atom("a", VAFSet((A,))), Lfc("a","b",abi.CMP_GREATER_EQUAL,0.5)])}))
def joint(bv):
    return run({"r": Conj([parse_formula("c:0.5"), Atom("a", VAFSet((A,))), Atom("b", VAFSet((bv,)))])})[1]
proj = A / math.sqrt(2.0)
def lse(v):
    m=max(v); return m if m==-math.inf else m+math.log(sum(math.exp(t-m) for t in v))
def integrate(include_end, res=0.2):
    lo, hi = 0.0, proj
    vis={}
    def f(p):
        ok = (p < proj) or (include_end and p == proj)
        vis[p] = joint(p) if ok else -math.inf
    L,R=lo,hi; f(L); f(R); first=None; mid=None
    while ((R-L)>=res and L<R) or mid is None:
        mid=(R+L)/2; f(mid); m1=(mid+L)/2; f(m1); m2=(R+mid)/2; f(m2)
        if first is None: first=mid
        xs=[L,m1,m2,R]; k=0
        for i in range(1,4):
            if vis[xs[i]]>vis[xs[k]]: k=i
        L,R = (xs[k-1] if k>0 else xs[k]), (xs[k+1] if k<3 else xs[k])
    f((first+hi)/2 if mid<first else (lo+first)/2)
    lo3=max(mid-3*res,lo); hi3=min(mid+3*res,hi); sa=(mid-lo3)/3; sb=(hi3-mid)/3
    for k in range(3): f(lo3+sa*k)
    for k in range(1,4): f(mid+sb*k)
    g=sorted(vis); terms=[]
    for a0,a1 in zip(g[:-1],g[1:]):
        w=(a1-a0)/2
        terms.append(lse([vis[a0],vis[a1]]) + (math.log(w) if w>0 else -math.inf))
    return lse(terms), [(p, vis[p]) for p in g]
for inc in (True, False):
    v, pts = integrate(inc)
    print("include_end", inc, "%.10f" % v)
    for p, q in pts: print("    %.17g %.6f" % (p, q))
print("proj", proj.hex())
