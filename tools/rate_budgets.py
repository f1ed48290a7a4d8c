"""Kernel ms of every BASELINE workload shape under each register budget (VLR_WAVES_PER_SIMD); honours VLR_LIB.
usage: python tools/rate_budgets.py [n_loci] [budgets, e.g. 2,3,4] [configs, e.g. config2,config3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from varlociraptor_amd import engine, synth
from bench import generate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
budgets = (sys.argv[2] if len(sys.argv) > 2 else "2,3,4").split(",")
names = (sys.argv[3] if len(sys.argv) > 3 else "config2,config3,config4,config5").split(",")
for name in names:
    cfg = synth.CONFIGS[name]()
    batch = generate(name, n, 0)
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    mo = int(batch.depth().sum(axis=1).max())
    for wpe in budgets:
        if wpe == "auto":
            os.environ.pop("VLR_WAVES_PER_SIMD", None)
        else:
            os.environ["VLR_WAVES_PER_SIMD"] = wpe
        plan = engine.Plan(cfg.scenario)
        plan.set_max_obs(mo)
        out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(4):
            plan.call_device(dbatch, out, st)
            torch.cuda.synchronize()
            ms.append(plan.last_kernel_ms())
        print("%s budget %s: %.2f ms = %.2f M loci/s (max_obs %d)" % (name, wpe, min(ms[1:]), n / min(ms[1:]) / 1e3, mo), flush=True)
        plan.close()
