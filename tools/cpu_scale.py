import sys, os, time, multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from varlociraptor_amd import synth
cfg = synth.config3(); b = synth.generate(cfg, 4096)
oracle.lib()
def work(rng):
    oracle.call(cfg.scenario, b, begin=rng[0], end=rng[1])
    return 0
if __name__ == "__main__":
    print("MALLOC_ARENA_MAX", os.environ.get("MALLOC_ARENA_MAX"))
    for procs in (1, 32, 128, 256):
        n = min(4096, procs * 16)
        bounds = np.linspace(0, n, procs + 1).astype(int)
        t0 = time.time()
        with mp.get_context("fork").Pool(procs) as pool:
            pool.map(work, [(int(bounds[i]), int(bounds[i + 1])) for i in range(procs)])
        dt = time.time() - t0
        print(procs, "processes", n, "loci", round(dt, 2), "s ->", round(n / dt, 1), "loci/s", flush=True)
