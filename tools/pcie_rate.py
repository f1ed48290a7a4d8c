"""PCIe-inclusive rate: host buffers in, host results out (vlr_batch_run_host), for DESIGN.md."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import engine, synth
cfg = synth.config3()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
from bench import generate
b = generate("config3", n, 0)
plan = engine.Plan(cfg.scenario)
plan.call_host(b.select(range(1000)))
t = time.perf_counter(); r = plan.call_host(b); dt = time.perf_counter() - t
print("host->device->host: %d loci in %.3f s = %.0f loci/s (kernel %.1f ms, %d MB in)" % (n, dt, n / dt, plan.last_kernel_ms(), b.algorithmic_bytes() >> 20))
t = time.perf_counter(); ra = plan.call_host(b, afd_capacity=96); dt2 = time.perf_counter() - t
print("with AFD replay: %.3f s = %.0f loci/s" % (dt2, n / dt2))
bp = engine.pin_batch(b)
plan.call_host(bp.select(range(1000)))
t = time.perf_counter(); r2 = plan.call_host(bp); dt3 = time.perf_counter() - t
import numpy as np
assert np.array_equal(r.ln_posterior, r2.ln_posterior, equal_nan=True)
print("page-locked input arrays (vlr_host_alloc): %.3f s = %.0f loci/s" % (dt3, n / dt3))
