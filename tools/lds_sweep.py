"""Same tumor-normal batch (depth 70), coefficient area padded to sweep the LDS footprint: kernel time per workgroups/CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from varlociraptor_amd import engine, synth
n = 100000
cfg = synth.config3(); cfg.depth = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
batch = synth.generate(cfg, n)
mo = int(batch.depth().sum(axis=1).max())
dbatch = engine.DeviceBatch(batch, "cuda:0")
os.environ["VLR_WAVES_PER_SIMD"] = sys.argv[1] if len(sys.argv) > 1 else "3"
for pad in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,20,40,60,80,100,120,140,160,180,200,240".split(","))]:
    plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo + pad)
    out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ms = []
    for i in range(3):
        plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
    print("max_obs %d (+%d): %.2f ms" % (mo + pad, pad, min(ms[1:])), flush=True)
    plan.close()
