"""Build-matrix probe: run a fixed set of seeded workloads through whichever engine build the environment selects
(VLR_LIB = path of a libvlr variant, VLR_WAVES_PER_SIMD = 1..4 picks the kernel's register budget) and dump the raw
results.  tests/test_gpu_build_matrix.py runs this once per (build, budget) in a subprocess and requires the dumps to be
bit-identical: a kernel whose results depend on the register allocator is wrong somewhere.

usage: python tools/matrix_run.py out.npz [quick|full]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def workloads(mode):
    from varlociraptor_amd import synth
    out = []

    def cfgd(cfg, depth=None, **kw):
        if depth is not None:
            cfg.depth = depth
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg

    c = cfgd(synth.config3(), 4.0, empty_fraction=0.25)
    out.append(("tn_tiny", c.scenario, synth.generate(c, 300, seed=12), 0))
    for d in (1.5, 3.0, 7.0, 12.0):
        c = cfgd(synth.config2(), d)
        out.append(("single_%g" % d, c.scenario, synth.generate(c, 300, seed=11), 0))
    c = synth.config3()
    out.append(("config3", c.scenario, synth.generate(c, 1500 if mode == "full" else 400, seed=3), 64))
    c = synth.config4()
    out.append(("config4", c.scenario, synth.generate(c, 400 if mode == "full" else 120, seed=4), 0))
    c = synth.config5()
    out.append(("config5", c.scenario, synth.generate(c, 600 if mode == "full" else 200, seed=5), 0))
    c = synth.config2()
    out.append(("config2", c.scenario, synth.generate(c, 2000 if mode == "full" else 500, seed=2), 64))
    # random scenarios of the fuzzer (same generator, seeded): nested ranges, sets, l2fc, contamination
    import fuzz_scenarios as fz
    from varlociraptor_amd import abi
    for seed, n_sc in ((1, 40 if mode == "full" else 12), (5, 40 if mode == "full" else 12)):
        rng = np.random.default_rng(seed)
        for it in range(n_sc):
            try:
                sc, names = fz.random_scenario(rng)
                sc.desc()
            except Exception:
                continue
            S = len(names)
            classes = []
            for _ in range(4):
                classes.append(("c", 0.25, tuple((float(v), float(v + w)) for v, w in zip(rng.choice([0.0, 0.1, 0.5, 1.0], S), rng.choice([0.0, 0.0, 0.2], S)))))
            classes = [(l, f, tuple((lo, min(hi, 1.0)) for lo, hi in spec)) for l, f, spec in classes]
            cfg = synth.SynthConfig(name="fuzz", config_id=50, scenario=sc, depth=float(rng.choice([4.0, 12.0, 30.0])),
                                    type_mix={abi.VT_SNV: 0.7, abi.VT_INDEL: 0.3}, classes=classes, purity=None)
            b = synth.generate(cfg, 24, seed=int(rng.integers(1 << 30)), bias_mask=abi.BIAS_ALL)
            out.append(("fuzz_%d_%d" % (seed, it), sc, b, 0))
    return out


def main():
    outp = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "quick"
    from varlociraptor_amd import engine
    res = {}
    for name, sc, batch, afd in workloads(mode):
        try:
            plan = engine.Plan(sc)
        except Exception as ex:  # the plan compiler rejects the scenario: identical for every build
            res[name + "/rejected"] = np.array([1])
            continue
        got = plan.call_host(batch, afd_capacity=afd)
        plan.close()
        for f in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status"):
            res[name + "/" + f] = np.asarray(getattr(got, f))
        if afd:
            res[name + "/afd_count"] = np.asarray(got.afd_count)
            cnt = np.minimum(np.asarray(got.afd_count), afd)
            mask = np.arange(afd)[None, None, :] < cnt[:, :, None]
            res[name + "/afd_vaf"] = np.where(mask, np.asarray(got.afd_vaf), 0.0)
            res[name + "/afd_lnprob"] = np.where(mask, np.asarray(got.afd_lnprob), 0.0)
    res["build_id"] = np.array([engine.build_id()])
    np.savez(outp, **res)
    print("matrix_run: %d arrays -> %s (lib %s, waves/SIMD %s)" % (len(res), outp, os.environ.get("VLR_LIB", "default"), os.environ.get("VLR_WAVES_PER_SIMD", "auto")))


def compare(paths, ignore_build_id=False):
    """Bitwise comparison of dumps against the first one; returns the number of differing arrays.  ignore_build_id: the dumps
    come from different sources on purpose (tools/compare_prev.sh: a restructured kernel against the previous commit's)."""
    base = np.load(paths[0])
    total = 0
    for p in paths[1:]:
        other = np.load(p)
        bad = []
        if not ignore_build_id and "build_id" in base.files and str(base["build_id"][0]) != str(other["build_id"][0]):
            print("%s was built from other sources (%s) than %s (%s): rebuild with varlociraptor_amd.engine.build_matrix()" %
                  (os.path.basename(p), other["build_id"][0], os.path.basename(paths[0]), base["build_id"][0]))
            total += 1
            continue
        for k in base.files:
            if ignore_build_id and k == "build_id":
                continue
            if k not in other.files:
                bad.append((k, "missing"))
                continue
            a, b = base[k], other[k]
            if a.shape != b.shape or not np.array_equal(a.view(np.uint8) if a.dtype.kind == "f" else a, b.view(np.uint8) if b.dtype.kind == "f" else b):
                if a.shape == b.shape and a.dtype.kind == "f":
                    with np.errstate(invalid="ignore"):
                        d = np.abs(a - b)
                    d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
                    nbad = int((np.nan_to_num(d, nan=np.inf) > 0).reshape(len(a), -1).any(axis=1).sum()) if a.ndim else int(d > 0)
                    bad.append((k, "max|d| %.3g, %d/%d rows" % (float(np.nan_to_num(d, nan=np.inf).max()), nbad, len(a) if a.ndim else 1)))
                else:
                    bad.append((k, "%d elements differ" % int((a != b).sum()) if a.shape == b.shape else "shape"))
        print("%s vs %s: %d of %d arrays differ" % (os.path.basename(p), os.path.basename(paths[0]), len(bad), len(base.files)))
        for k, why in bad[:12]:
            print("    ", k, why)
        total += len(bad)
    return total


def compare_tol(paths, tol=1e-9):
    """Tolerance comparison (probability space) of dumps against the first one: for changes of the arithmetic order."""
    base = np.load(paths[0])
    worst = 0
    for p in paths[1:]:
        other = np.load(p)
        nbad = 0
        for k in base.files:
            if k == "build_id":
                continue
            if k.endswith("/rejected") or k not in other.files or "/afd_" in k:
                continue
            a, b = base[k], other[k]
            name = k.split("/")[1]
            if name in ("ln_posterior",):
                with np.errstate(invalid="ignore", over="ignore"):
                    d = np.abs(np.exp(a) - np.exp(b))
                d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
                m = float(np.nan_to_num(d, nan=np.inf).max()) if d.size else 0.0
                rows = np.nonzero(np.nan_to_num(d, nan=np.inf).reshape(len(a), -1).max(axis=1) > tol)[0]
                if len(rows):
                    nbad += 1
                    print("    %s: max |dP| %.3g, rows %s" % (k, m, rows[:8].tolist()))
            elif name in ("map_vaf",):
                d = np.where(np.isnan(a) & np.isnan(b), 0.0, np.abs(a - b))
                rows = np.nonzero(np.nan_to_num(d, nan=np.inf).reshape(len(a), -1).max(axis=1) > 0)[0]
                if len(rows):
                    nbad += 1
                    print("    %s: %d rows with another MAP VAF %s" % (k, len(rows), rows[:8].tolist()))
            elif name in ("map_bias", "best_event", "status"):
                if not np.array_equal(a, b):
                    nbad += 1
                    print("    %s: %d elements differ" % (k, int((a != b).sum())))
        print("%s vs %s (tol %g): %d arrays outside" % (os.path.basename(p), os.path.basename(paths[0]), tol, nbad))
        worst += nbad
    return worst


if __name__ == "__main__":
    if sys.argv[1] == "compare":
        sys.exit(1 if compare(sys.argv[2:]) else 0)
    if sys.argv[1] == "compare-tol":
        sys.exit(1 if compare_tol(sys.argv[2:]) else 0)
    main()
