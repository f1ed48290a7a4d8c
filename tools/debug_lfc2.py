import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Sample, Scenario, Conj, Atom, Lfc, VAFSet
sc0=Scenario({"a":Sample(resolution=0.02, universe="[0.0,1.0]"), "b":Sample(resolution=0.1, universe="[0.0,1.0]")}, {"e":"a:0.5"})
cfg=synth.SynthConfig(name="t", config_id=52, scenario=sc0, depth=30.0, type_mix={abi.VT_SNV:0.7, abi.VT_INDEL:0.3}, classes=[("c",0.5,((0.0,0.0),(0.0,0.0))),("d",0.5,((0.1,0.3),(0.0,0.2)))])
b=synth.generate(cfg, 200, seed=5)
sub=b.select([53])
x = 0.019736842105263157
proj = float.fromhex("0x1.c94fdfabba9d9p-6")
for X in (proj, np.nextafter(proj, 0), np.nextafter(proj, 1)):
    ev = {"r1": Conj([Lfc("a","b",abi.CMP_LESS,0.5), Atom("a", VAFSet((float(X),))), Atom("b", VAFSet((x,)))])}
    sc=Scenario(sc0.samples, ev)
    plan=engine.Plan(sc); g=plan.call_host(sub); plan.close()
    r=oracle.call(sc,sub,want_events=True)
    print(float(X).hex(), "gpu", g.ln_posterior[0,1], "ref", r.ln_posterior[0,1])
from varlociraptor_amd.scenario import VAFRange
def val(ev):
    sc=Scenario(sc0.samples, ev)
    plan=engine.Plan(sc); g=plan.call_host(sub); plan.close()
    r=oracle.call(sc,sub,want_events=True)
    return g.ln_posterior[0,1]+g.ln_marginal[0], r.ln_posterior[0,1]+r.ln_marginal[0]
print("lfc only        ", val({"r1": Conj([Lfc("a","b",abi.CMP_LESS,0.5), Atom("b", VAFSet((x,)))])}))
print("lfc + a:[0,proj]", val({"r1": Conj([Lfc("a","b",abi.CMP_LESS,0.5), Atom("a", VAFRange(0.0, proj, False, False)), Atom("b", VAFSet((x,)))])}))
print("a:[0,proj] no lfc", val({"r1": Conj([Atom("a", VAFRange(0.0, proj, False, False)), Atom("b", VAFSet((x,)))])}))
print("a:[0,proj[ no lfc", val({"r1": Conj([Atom("a", VAFRange(0.0, proj, False, True)), Atom("b", VAFSet((x,)))])}))
