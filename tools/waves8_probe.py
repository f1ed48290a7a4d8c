"""Do the 6- and 8-wave instances of the call kernel (80 / 64 VGPRs, the build matrix's `stress` library) pay on shallow workloads whose
LDS footprint allows 24-32 workgroups per CU (round 6)?   VLR_LIB=.../libvlr_stress.so python tools/waves8_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from varlociraptor_amd import engine, synth
n = 200000
cases = [("config2", synth.config2())]
c = synth.config2(); c.depth = 10.0
cases.append(("config2 depth 10", c))
c = synth.config3(); c.depth = 15.0
cases.append(("config3 depth 15", c))
for name, cfg in cases:
    batch = synth.generate(cfg, n)
    mo = int(batch.depth().sum(axis=1).max())
    dbatch = engine.DeviceBatch(batch, "cuda:0")
    res = {}
    for wpe in ("3", "4", "6", "8"):
        os.environ["VLR_WAVES_PER_SIMD"] = wpe
        os.environ["VLR_DEBUG_LAUNCH"] = "1" if wpe == "8" else ""
        if not os.environ["VLR_DEBUG_LAUNCH"]:
            del os.environ["VLR_DEBUG_LAUNCH"]
        plan = engine.Plan(cfg.scenario); plan.set_max_obs(mo)
        out = engine.DeviceResults(batch.n_loci, plan.n_out, plan.n_samples, "cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        ms = []
        for i in range(4):
            plan.call_device(dbatch, out, st); torch.cuda.synchronize(); ms.append(plan.last_kernel_ms())
        res[wpe] = min(ms[1:]); plan.close()
    print("%s max_obs %d: " % (name, mo) + ", ".join("%s waves %.2f ms" % (w, res[w]) for w in ("3", "4", "6", "8")), flush=True)
