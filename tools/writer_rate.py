"""CPU-only rate of the native calls writer: a synthetic config3 table (host reader) + synthetic results with AFD lists -> calls BCF.
   usage: python tools/writer_rate.py [records] [afd entries per list]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import callsfmt, ingest, synth
from varlociraptor_amd.batch import CallResults
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = synth.config3()
b = synth.generate(cfg, n, seed=1)
tmp = tempfile.mkdtemp()
paths = []
for s, name in enumerate(cfg.scenario.sample_names):
    p = os.path.join(tmp, name + ".bcf"); ingest.write_observations(p, b, s); paths.append(p)
batch, sites = ingest.read_observations(paths)
table = batch.extra["native_table"]
sc = cfg.scenario
names = sc.out_names()
rng = np.random.default_rng(3)
res = CallResults(n, len(names), len(sc.sample_names), 96)
lp = rng.normal(-3, 2, (n, len(names))); lp -= np.logaddexp.reduce(lp, axis=1)[:, None]
res.ln_posterior[:] = lp; res.ln_marginal[:] = -100.0
res.map_vaf[:] = rng.random((n, len(sc.sample_names)))
res.status[:] = 0; res.best_event[:] = 0; res.map_bias[:] = 0
res.afd_count[:] = k
res.afd_vaf[:, :, :k] = np.sort(rng.random((n, len(sc.sample_names), k)), axis=2)
res.afd_lnprob[:, :, :k] = -rng.random((n, len(sc.sample_names), k)) * 30
header = callsfmt.header(names, sc.sample_names, list(sites.contig_names))
out = os.path.join(tmp, "calls.bcf")
for rep in range(3):
    t0 = time.perf_counter()
    ingest.write_calls(out, header, table, res, names)
    dt = time.perf_counter() - t0
    t = ingest.last_timings()
    print("%d records in %.3f s = %.0f records/s; encode %.3f deflate+write %.3f; %d bytes" % (n, dt, n / dt, t["encode"], t["deflate_write"], os.path.getsize(out)))
import hashlib
print("sha1", hashlib.sha1(open(out, "rb").read()).hexdigest())
