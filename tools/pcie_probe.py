"""Which ingredient of bench.py's pcie_inclusive leg costs a third of its rate?  (round 6)"""
import sys, os, time
sys.path.insert(0, os.getcwd())
from varlociraptor_amd import engine, synth
from varlociraptor_amd.batch import CallResults
from bench import generate
cfg = synth.config3(); n = 1000000
b = generate("config3", n, 0)
bp = engine.pin_batch(b)
plan = engine.Plan(cfg.scenario)
def t(label, **kw):
    t0 = time.perf_counter(); plan.call_host(bp, **kw); dt = time.perf_counter() - t0
    print("%-50s %.3f s = %.2f M loci/s" % (label, dt, n / dt / 1e6), flush=True)
r = CallResults(n, plan.n_out, plan.n_samples, 0, alloc=engine.host_array)
plan.call_host(bp); plan.call_host(bp, results=r)
for i in range(3):
    t("default results"); t("page-locked results", results=r)
import torch
x = torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
for i in range(2):
    t("torch initialised: default results"); t("torch initialised: page-locked results", results=r)
big = torch.empty(8 << 30, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
for i in range(2):
    t("8 GiB held by torch: default results"); t("8 GiB held by torch: page-locked results", results=r)
del big; torch.cuda.empty_cache()
for i in range(2):
    t("released: default results"); t("released: page-locked results", results=r)
