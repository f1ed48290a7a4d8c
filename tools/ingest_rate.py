"""CPU-only rate of the native streaming reader (no GPU needed): synthetic config3 observation BCFs -> vlr_obs_reader chunks.
   usage: python tools/ingest_rate.py [records] [chunk]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import ingest, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg = synth.config3()
b = synth.generate(cfg, n, seed=1)
tmp = tempfile.mkdtemp()
paths = []
for s, name in enumerate(cfg.scenario.sample_names):
    p = os.path.join(tmp, name + ".bcf"); ingest.write_observations(p, b, s); paths.append(p)
for rep in range(3):
    ingest.total_timings(reset=True)
    t0 = time.perf_counter()
    r = ingest.ObsReader(paths, chunk_records=chunk)
    k = 0
    while True:
        it = r.next()
        if it is None: break
        k += it[0].n_loci
    r.close()
    dt = time.perf_counter() - t0
    t = ingest.total_timings()
    print("%d records in %.3f s = %.0f records/s; inflate %.3f (summed over files) files_wall %.3f merge %.3f" % (k, dt, k / dt, t["inflate"], t["files_wall"], t["merge"]))
