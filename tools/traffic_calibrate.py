"""Streams of known size for the FETCH_SIZE / WRITE_SIZE calibration (run under rocprofv3 --pmc, see tools/traffic_measure.sh).
usage: python tools/traffic_calibrate.py read|write [n_elements] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import engine
mode = 0 if sys.argv[1] == "read" else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 30)   # 4 GiB read / 8 GiB written: far above the 256 MiB Infinity Cache
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = engine.lib()
L.vlr_selftest_stream.restype = C.c_int
L.vlr_selftest_stream.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int]
rc = L.vlr_selftest_stream(0, mode, n, reps)
assert rc == 0, L.vlr_last_error()
print("stream %s: %d launches of %d bytes" % (sys.argv[1], reps, n * (4 if mode == 0 else 8)))
