"""Per-phase wave-cycle breakdown (needs a -DVLR_PROFILE build: VLR_LIB=variants/libvlr_prof.so)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from varlociraptor_amd import engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
cfg = synth.CONFIGS[name]()
b = synth.generate(cfg, n)
plan = engine.Plan(cfg.scenario)
plan.call_host(b)
L = engine.lib()
out = (C.c_ulonglong * 40)()
L.vlr_plan_profile_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert L.vlr_plan_profile_counters(plan._h, out) == 0
names = ["A stats", "gating", "coefficients", "walk/other", "deliver held/deferred", "single integrate", "batch prep", "batch rounds", "batch epilogue", "phase C", "#batch runs", "#single chains", "round: products", "round: reduce", "round: log+prior", "round: advance", "outer: task setup", "outer: vary eval", "outer: entry/delivery", "iter: next root", "iter: root entry", "walk", "iter: root exit", "discrete roots", "walk: resume", "deliver: fetch", "deliver: MAP", "outer: advance", "outer: finish", "outer: issue", "outer: begin", "loop: pre-batch", "walk: sample node", "walk: range setup", "walk: deferred leaf", "walk: to leaf"] + ["-"] * 4
cnt = {10, 11}
tot = sum(v for i, v in enumerate(out) if i not in cnt)
for nm, v in zip(names, out):
    print("%-18s %14d  %5.1f%%  per locus %9.0f" % (nm, v, 100.0 * v / max(tot, 1), v / n))
print("batch runs %.1f, single chains %.1f, batch rounds %.1f, active rows per round %.2f" % ((out[10] & 0xffffffff) / n, (out[11] & 0xffffffff) / n, (out[10] >> 32) / n, (out[11] >> 32) / max(out[10] >> 32, 1)))
print("evals, terms", plan.work_counters())
