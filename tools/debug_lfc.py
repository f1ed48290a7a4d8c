import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from parity import compare, describe
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Sample, Scenario, Contamination
def run(events, cont, res_a=0.02, res_b=0.1, depth=30.0, n=200, label=""):
    sc=Scenario({"a":Sample(resolution=res_a, universe="[0.0,1.0]", contamination=Contamination("b",cont) if cont else None), "b":Sample(resolution=res_b, universe="[0.0,1.0]")}, events)
    cfg=synth.SynthConfig(name="t", config_id=52, scenario=sc, depth=depth, type_mix={abi.VT_SNV:0.7, abi.VT_INDEL:0.3}, classes=[("c",0.5,((0.0,0.0),(0.0,0.0))),("d",0.5,((0.1,0.3),(0.0,0.2)))])
    b=synth.generate(cfg, n, seed=5)
    plan=engine.Plan(sc); got=plan.call_host(b); plan.close()
    ref=oracle.call(sc,b,want_events=True)
    m=compare(got,ref,label=label+str(events))
    print(describe(m))
    return m, got, ref, b
run({"r1":"b:[0.0,0.5[ & l2fc(a,b) < 0.5"}, 0.1)
run({"r1":"b:[0.0,0.5[ & l2fc(a,b) < 0.5"}, None)
run({"r1":"b:[0.0,0.5[ & l2fc(a,b) < 1.0"}, None)
run({"r1":"b:[0.0,0.5[ & l2fc(a,b) <= 0.5"}, None)
run({"r1":"b:[0.0,0.5[ & l2fc(a,b) < 0.5"}, None, res_a=0.1)
run({"r1":"b:]0.0,0.5[ & l2fc(a,b) < 0.5"}, None)
run({"r1":"b:[0.1,0.4] & l2fc(a,b) < 0.5"}, None)
print("---- discrete b")
for bv in ["0.005","0.0125","0.02","0.03","0.05","0.1","0.2","0.3","0.45"]:
    run({"r1":"b:%s & l2fc(a,b) < 0.5" % bv}, None, label="b=%s " % bv)
print("---- details")
m, got, ref, b = run({"r1":"b:[0.0,0.5[ & l2fc(a,b) < 0.5"}, None)
np.set_printoptions(precision=7, linewidth=200)
for l in m["bad"][:6]:
    sl_a, sl_b = b.pileup_slice(l,0), b.pileup_slice(l,1)
    pa, pr = b.columns["prob_alt"], b.columns["prob_ref"]
    print("locus", l, "depth", b.depth()[l], "status %x" % got.status[l], "n_alt a", int((pa[sl_a]>pr[sl_a]).sum()), "n_alt b", int((pa[sl_b]>pr[sl_b]).sum()))
    print("  got", got.ln_posterior[l], got.ln_marginal[l], " ref", ref.ln_posterior[l], ref.ln_marginal[l])
ok=[l for l in range(200) if l not in m["bad"]][:6]
for l in ok:
    sl_a, sl_b = b.pileup_slice(l,0), b.pileup_slice(l,1)
    pa, pr = b.columns["prob_alt"], b.columns["prob_ref"]
    print("ok locus", l, "depth", b.depth()[l], "n_alt a", int((pa[sl_a]>pr[sl_a]).sum()), "n_alt b", int((pa[sl_b]>pr[sl_b]).sum()))
print("---- exact outer points for locus 53")
from varlociraptor_amd.scenario import Conj, Atom, Lfc, VAFSet
sub = b.select([53])
hi = 18.0/38.0
pts = [0.0, hi]
L, R = 0.0, hi
for it in range(8):
    mid=(R+L)/2; m1=(mid+L)/2; m2=(R+mid)/2
    pts += [mid, m1, m2]
    R = m1  # bracket towards 0 (all-ref locus)
    if R - L < 0.1: break
lo3=max(mid-0.3,0.0); hi3=min(mid+0.3,hi); sa=(mid-lo3)/3.0; sb=(hi3-mid)/3.0
pts += [lo3+sa*0.0, lo3+sa*1.0, lo3+sa*2.0, mid+sb*1.0, mid+sb*2.0, mid+sb*3.0]
for x in pts:
    ev = {"r1": Conj([Lfc("a","b",abi.CMP_LESS,0.5), Atom("b", VAFSet((x,)))])}
    sc=Scenario({"a":Sample(resolution=0.02, universe="[0.0,1.0]"), "b":Sample(resolution=0.1, universe="[0.0,1.0]")}, ev)
    plan=engine.Plan(sc); g=plan.call_host(sub); plan.close()
    r=oracle.call(sc,sub,want_events=True)
    gv = g.ln_posterior[0,1]+g.ln_marginal[0]; rv = r.ln_posterior[0,1]+r.ln_marginal[0]
    print("  b=%.17g gpu %.12f ref %.12f diff %.3g" % (x, gv, rv, gv-rv))
print("---- AFD at b=0.019736842105263157")
x = 0.019736842105263157
ev = {"r1": Conj([Lfc("a","b",abi.CMP_LESS,0.5), Atom("b", VAFSet((x,)))])}
sc=Scenario({"a":Sample(resolution=0.02, universe="[0.0,1.0]"), "b":Sample(resolution=0.1, universe="[0.0,1.0]")}, ev)
plan=engine.Plan(sc); g=plan.call_host(sub, afd_capacity=64); plan.close()
r=oracle.call(sc,sub,afd_capacity=64,want_events=True)
print("best", g.best_event, r.best_event, "map", g.map_vaf, r.map_vaf)
for name, o in (("gpu", g), ("ref", r)):
    n=int(o.afd_count[0,0]); idx=np.argsort(o.afd_vaf[0,0,:n])
    print(name, n)
    for i in idx: print("   %.17g  %.10f" % (o.afd_vaf[0,0,i], o.afd_lnprob[0,0,i] + o.ln_marginal[0]))
print("proj", x / 0.7071067811865476)
print("---- python emulation")
import math
def joint(a):
    ev = {"r": Conj([Atom("a", VAFSet((a,))), Atom("b", VAFSet((x,)))])}
    sc2=Scenario({"a":Sample(resolution=0.02, universe="[0.0,1.0]"), "b":Sample(resolution=0.1, universe="[0.0,1.0]")}, ev)
    r2=oracle.call(sc2,sub,want_events=True)
    return r2.ln_posterior[0,1]+r2.ln_marginal[0]
proj = x / math.sqrt(0.5)  # approx
import struct
def lse(v):
    m=max(v); 
    return m if m==-math.inf else m+math.log(sum(math.exp(t-m) for t in v))
def integrate(include_end):
    lo, hi, res = 0.0, proj, 0.02
    vis={}
    def f(p):
        ok = (p < proj) or (include_end and p == proj)
        v = joint(p) if ok else -math.inf
        vis[p]=v; return v
    L,R=lo,hi; f(L); f(R); first=None; mid=None
    while ((R-L)>=res and L<R) or mid is None:
        mid=(R+L)/2; f(mid); m1=(mid+L)/2; f(m1); m2=(R+mid)/2; f(m2)
        if first is None: first=mid
        xs=[L,m1,m2,R]; k=0
        for i in range(1,4):
            if vis[xs[i]]>vis[xs[k]]: k=i
        L2 = xs[k-1] if k>0 else xs[k]; R2 = xs[k+1] if k<3 else xs[k]; L,R=L2,R2
    f((first+hi)/2 if mid<first else (lo+first)/2)
    lo3=max(mid-3*res,lo); hi3=min(mid+3*res,hi); sa=(mid-lo3)/3; sb=(hi3-mid)/3
    for k in range(3): f(lo3+sa*k)
    for k in range(1,4): f(mid+sb*k)
    g=sorted(vis)
    terms=[]
    for a0,a1 in zip(g[:-1],g[1:]):
        w=(a1-a0)/2
        va,vb=vis[a0],vis[a1]
        t=lse([va,vb]) + (math.log(w) if w>0 else -math.inf)
        terms.append(t)
    return lse(terms), g
for inc in (False, True):
    val, g = integrate(inc)
    print("include_end", inc, "integral %.12f" % val, "points", len(g))
    print("   ", ["%.6g"%p for p in g])
