import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from varlociraptor_amd import engine, synth
from bench import generate
n=200000
for name in os.environ.get("VLR_RATE_CONFIGS", "config3,config2").split(","):
    cfg = synth.CONFIGS[name]()
    batch = generate(name, n, 0)
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    plan = engine.Plan(cfg.scenario)
    try:
        plan.fit_max_obs(batch.obs_offset)
    except AttributeError:   # a library built before vlr_plan_fit_max_obs existed
        plan.set_max_obs(int(batch.depth().sum(axis=1).max()))
    out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ms=[]
    for i in range(4):
        plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
    plan.work_counters(reset=True)
    plan.call_device(dbatch, out, st); torch.cuda.synchronize()
    ev, terms = plan.work_counters()
    print(os.environ.get("VLR_LIB","default").split("/")[-1], name, "%.2f ms" % min(ms[1:]), "evals/locus %.1f terms/locus %.0f" % (ev / n, terms / n), flush=True)
