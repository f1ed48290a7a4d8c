import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from oracle import oracle
from parity import compare, describe
from varlociraptor_amd import abi, engine, synth
from varlociraptor_amd.scenario import Sample, Scenario
def run(events, depth, S=3, res=0.1, n=24):
    names=["a","b","c"][:S]
    sc=Scenario({k:Sample(resolution=res, universe="[0.0,1.0]") for k in names}, events)
    cfg=synth.SynthConfig(name="t", config_id=51, scenario=sc, depth=depth, type_mix={abi.VT_SNV:1.0}, classes=[("c",1.0,tuple((0.0,0.5) for _ in names))])
    b=synth.generate(cfg, n, seed=3)
    plan=engine.Plan(sc); got=plan.call_host(b); plan.close()
    ref=oracle.call(sc,b,want_events=True)
    m=compare(got,ref,label=str(events)+" depth %g"%depth)
    print(describe(m), "statuses", sorted(set(int(x) for x in got.status)))
run({"e":"c:[0.0,1.0]"}, 3.0)
run({"e":"c:[0.0,1.0]"}, 8.0)
run({"e":"c:[0.0,1.0]"}, 20.0)
run({"e":"a:{0.5,1.0}"}, 3.0)
run({"e":"a:{0.5,1.0}"}, 20.0)
run({"e":"b:[0.0,1.0]"}, 3.0, S=2)
run({"e":"b:[0.0,1.0]"}, 8.0, S=2)
