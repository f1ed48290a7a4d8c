"""What the host side of the GPU box offers: visible hardware threads, affinity, cgroup quota, and how a GIL-free workload
(zlib.crc32 over private buffers) scales with the thread count.  Sets the expectations for the ingest and CPU-baseline numbers."""
import os
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.strerror)
try:
    print([l.strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0])
    print([l.strip() for l in open("/proc/meminfo")][:3])
except Exception:
    pass
buf = os.urandom(1 << 24)


def work(_):
    c = 0
    for _ in range(24):
        c = zlib.crc32(buf, c)
    return c


base = None
for t in (1, 4, 8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=t) as ex:
        list(ex.map(work, range(t)))
    dt = time.perf_counter() - t0
    base = base or dt
    print("threads %3d: %.2f s for %d units -> speed-up %.1f" % (t, dt, t, t * base / dt))
