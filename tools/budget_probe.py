"""Kernel + deep-launch time of config 3 (200 000 loci, 100x) under LDS budgets around the 16-workgroup boundary (round 6)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from varlociraptor_amd import engine, synth
from bench import generate
n = 200000
cfg = synth.CONFIGS["config3"]()
batch = generate("config3", n, 0)
d = batch.depth().sum(axis=1)
dbatch = engine.DeviceBatch(batch, "cuda:0")
for mo in [int(x) for x in os.environ.get('VLR_PROBE_BUDGETS','266,243,242,241,240').split(',')]:
    plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo)
    out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ts, ms = [], []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.call_device(dbatch, out, st); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3); ms.append(plan.last_kernel_ms())
    bad = int((out.status.cpu().numpy() & 8).sum())
    print("max_obs %d: %.2f%% of the loci above; call kernel %.2f ms, whole step %.2f ms, still flagged too deep: %d" % (mo, 100.0 * (d > mo).mean(), min(ms[1:]), min(ts[1:]), bad), flush=True)
    plan.close()
plan = engine.Plan(cfg.scenario)
print("vlr_plan_fit_max_obs ->", plan.fit_max_obs(batch.obs_offset), flush=True)
out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
st = torch.cuda.current_stream().cuda_stream
ts = []
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.call_device(dbatch, out, st); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("fitted budget: steps", ["%.2f" % t for t in ts], flush=True)
