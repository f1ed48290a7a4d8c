"""Is the gain of the 4-wave build a matter of the pileup depth or of the workgroup count (round 6)?  Tumor-normal at 30x with the
coefficient area padded: 16, 15, 14, 13, 12 workgroups per CU under 3 and 4 waves per SIMD."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from varlociraptor_amd import engine, synth
n = 100000
cfg = synth.config3(); cfg.depth = 30.0
batch = synth.generate(cfg, n)
mo = int(batch.depth().sum(axis=1).max())
dbatch = engine.DeviceBatch(batch, "cuda:0")
for pad in (0, 24, 60, 104, 150, 210):
    res = {}
    for wpe in ("3", "4"):
        os.environ["VLR_WAVES_PER_SIMD"] = wpe
        plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo + pad)
        out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(4):
            plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
        res[wpe] = min(ms[1:]); plan.close()
    lds = 8568 + 16 * (mo + pad)   # bytes per workgroup of this scenario: static + tables + 16 B per observation slot (see call_kernel_dyn_lds)
    print("max_obs %d (+%d) ~%d B -> %d workgroups/CU: 3 waves %.2f ms, 4 waves %.2f ms (ratio %.3f)" % (mo + pad, pad, lds, 163840 // ((lds + 511) // 512 * 512), res["3"], res["4"], res["3"] / res["4"]), flush=True)
