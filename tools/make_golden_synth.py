"""Golden vectors of the CPU oracle on seeded synthetic loci of BASELINE configs 2-5 (VERDICT r1 #6b):
per-event ln posteriors, marginal, MAP VAFs, bias codes, best event, status for 1000 loci per config and the visited-point
(AFD) lists of the first 200.  A silent change of the oracle itself shows up in tests/test_golden_synth.py; the GPU suite
compares the engine with the same files.  The inputs are regenerated from the seed (varlociraptor_amd.synth); a digest of
the input columns is stored to detect generator drift.

usage: python tools/make_golden_synth.py        (writes tests/golden/synth/config{2,3,4,5}.npz)"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

N_LOCI, N_AFD, AFD_CAP, SEED = 1000, 200, 128, 20260928


def inputs(name):
    from varlociraptor_amd import synth
    cfg = synth.CONFIGS[name]()
    return cfg, synth.generate(cfg, N_LOCI, seed=SEED + cfg.config_id)


def digest(batch):
    h = hashlib.sha1()
    h.update(batch.obs_offset.tobytes())
    for k in sorted(batch.columns):
        h.update(np.ascontiguousarray(batch.columns[k]).tobytes())
    for k in sorted(batch.locus):
        h.update(np.ascontiguousarray(batch.locus[k]).tobytes())
    return h.hexdigest()


def evaluate(cfg, batch):
    from oracle import oracle
    res = oracle.call(cfg.scenario, batch, want_events=True)
    afd = oracle.call(cfg.scenario, batch.select(np.arange(N_AFD)), afd_capacity=AFD_CAP)
    out = {f: np.asarray(getattr(res, f)) for f in ("ln_posterior", "ln_marginal", "map_vaf", "map_bias", "best_event", "status")}
    out["event_ln_posterior"] = res.event_ln_posterior
    cnt = np.minimum(afd.afd_count, AFD_CAP)
    mask = np.arange(AFD_CAP)[None, None, :] < cnt[:, :, None]
    out["afd_count"] = afd.afd_count
    out["afd_vaf"] = np.where(mask, afd.afd_vaf, 0.0)
    out["afd_lnprob"] = np.where(mask, afd.afd_lnprob, 0.0)
    return out


def main():
    d = os.path.join(ROOT, "tests", "golden", "synth")
    os.makedirs(d, exist_ok=True)
    for name in ("config2", "config3", "config4", "config5"):
        cfg, batch = inputs(name)
        out = evaluate(cfg, batch)
        out["input_digest"] = np.array([digest(batch)])
        out["out_names"] = np.array(cfg.scenario.out_names())
        np.savez_compressed(os.path.join(d, name + ".npz"), **out)
        print(name, batch.n_loci, "loci", batch.n_obs, "observations ->", os.path.join(d, name + ".npz"))


if __name__ == "__main__":
    main()
