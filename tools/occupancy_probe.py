"""2- vs 3-waves-per-SIMD kernel builds on tumor-normal batches of different depth (LDS footprint), for the launcher's threshold."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from varlociraptor_amd import engine, synth
n = 100000
for depth in [30.0, 45.0, 55.0, 70.0, 100.0]:
    cfg = synth.config3(); cfg.depth = depth
    batch = synth.generate(cfg, n)
    mo = int(batch.depth().sum(axis=1).max())
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    res = {}
    for wpe in ("2", "3"):
        os.environ["VLR_WAVES_PER_SIMD"] = wpe
        plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo)
        out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(3):
            plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
        res[wpe] = min(ms[1:]); plan.close()
    print("depth %.0f max_obs %d: 2 waves %.2f ms, 3 waves %.2f ms (ratio %.3f)" % (depth, mo, res["2"], res["3"], res["2"] / res["3"]), flush=True)
