"""HBM-resident kernel rate of every BASELINE workload shape (reduced locus counts), for DESIGN.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from varlociraptor_amd import engine, synth
from bench import generate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for name in ["config2", "config3", "config4", "config5"]:
    cfg = synth.CONFIGS[name]()
    batch = generate(name, n, 0)
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    plan = engine.Plan(cfg.scenario)
    plan.set_max_obs(int(batch.depth().sum(axis=1).max()))
    out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    plan.call_device(dbatch, out, st); torch.cuda.synchronize()
    plan.work_counters(reset=True)
    plan.call_device(dbatch, out, st); torch.cuda.synchronize()
    ms = plan.last_kernel_ms()
    ev, terms = plan.work_counters()
    res = out.to_host()
    stat = np.bincount(res.status & 0xF, minlength=16)
    print("%s: %d loci, kernel %.2f ms = %.0f loci/s; %.0f pileup evaluations and %.0f terms per locus; error bits %s" %
          (name, n, ms, n / ms * 1e3, ev / n, terms / n, {i: int(v) for i, v in enumerate(stat) if v and i}), flush=True)
    plan.close()
