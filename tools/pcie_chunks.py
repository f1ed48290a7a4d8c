import sys, os, time
sys.path.insert(0, os.getcwd())
from varlociraptor_amd import engine, synth
from bench import generate
cfg = synth.config3()
n = 1000000
b = generate("config3", n, 0)
bp = engine.pin_batch(b)
plan = engine.Plan(cfg.scenario)
plan.call_host(b.select(range(1000)))
for mb in [64, 256, 512, 1024, 2048]:
    os.environ["VLR_HOST_CHUNK_MB"] = str(mb)
    for name, bb in (("pageable", b), ("pinned", bp)):
        plan.call_host(bb)
        t = time.perf_counter(); plan.call_host(bb); dt = time.perf_counter() - t
        print("chunk %4d MB %-8s: %.3f s = %.0f loci/s" % (mb, name, dt, n / dt), flush=True)
