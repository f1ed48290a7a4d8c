"""Rate of the native streaming reader alone: synthetic config3 observation BCFs -> vlr_obs_reader chunks.  Host reader (no GPU
   needed) or, with a third argument `device`, the device reader (inflate, record split and v15 decode as kernels).
   usage: python tools/ingest_rate.py [records] [chunk] [device]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import ingest, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = 0 if len(sys.argv) > 3 and sys.argv[3] == "device" else None
cfg = synth.config3()
b = synth.generate(cfg, n, seed=1)
tmp = tempfile.mkdtemp()
paths = []
for s, name in enumerate(cfg.scenario.sample_names):
    p = os.path.join(tmp, name + ".bcf"); ingest.write_observations(p, b, s); paths.append(p)
for rep in range(3):
    ingest.total_timings(reset=True)
    t0 = time.perf_counter()
    if dev is not None: ingest.device_timings(reset=True)
    r = ingest.ObsReader(paths, chunk_records=chunk, device=dev)
    k = 0
    while True:
        it = r.next()
        if it is None: break
        k += it[0].n_loci
    r.close()
    dt = time.perf_counter() - t0
    t = ingest.total_timings()
    if dev is not None:
        d = ingest.device_timings()
        print("device reader: %d records in %.3f s = %.0f records/s; " % (k, dt, k / dt) + ", ".join("%s %.3f" % (a, d[a]) for a in ("feed_inflate", "split_scan", "decode", "copy_back", "host_table", "inflate_kernel", "decode_wait", "total")) +
              "; %.2f GB inflated from %.2f GB, %d serial walks" % (d["inflated_bytes"] / 1e9, d["compressed_bytes"] / 1e9, d["serial_walks"]))
        continue
    print("%d records in %.3f s = %.0f records/s; inflate %.3f (summed over files) files_wall %.3f merge %.3f" % (k, dt, k / dt, t["inflate"], t["files_wall"], t["merge"]))
