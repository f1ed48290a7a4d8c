"""3 vs 4 waves per SIMD of the call kernel where the LDS footprint allows 16 workgroups per CU (round 6): shallow tumor-normal batches,
config 2 (single sample, 30x) and config 5 (four samples, 60x)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from varlociraptor_amd import engine, synth
n = 100000
cases = []
for depth in (20.0, 30.0, 45.0):
    cfg = synth.config3(); cfg.depth = depth
    cases.append(("config3 depth %.0f" % depth, cfg))
cases.append(("config2", synth.config2()))
cases.append(("config5", synth.config5()))
for name, cfg in cases:
    batch = synth.generate(cfg, n)
    mo = int(batch.depth().sum(axis=1).max())
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    res = {}
    for wpe in ("3", "4"):
        os.environ["VLR_WAVES_PER_SIMD"] = wpe
        plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo)
        out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(4):
            plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
        res[wpe] = min(ms[1:]); plan.close()
    print("%s max_obs %d: 3 waves %.2f ms, 4 waves %.2f ms (ratio %.3f)" % (name, mo, res["3"], res["4"], res["3"] / res["4"]), flush=True)
