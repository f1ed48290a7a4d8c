"""How far can the exact pair-HMM restatement (oracle/vlr_realign_oracle.cpp, what the GPU kernel is compared with) be from the
real `bio::stats::pairhmm` crate?  The three crate behaviours that cannot be verified in this image are switched on one by one
and together; reported: max |delta| of the NORMALISED allele supports (what ends up in prob_ref / prob_alt,
realignment/mod.rs:359-385) in probability space over the bench workload's read/allele pairs.  CPU only.
usage: python tools/realign_crate_bound.py [n_reads] [seed] > profiles/r03_realign_crate_bound.json"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle
from varlociraptor_amd import realign, realign_synth


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pb, _ = realign_synth.generate(n_reads, seed=seed)
    gap = realign.GapParams()
    g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
    oracle.lib()
    n = len(pb)

    def run(bits):
        f = lambda k: oracle.pairhmm_prob_related_variant(pb.x[k], pb.y[k], pb.q[k], g, pb.band[k], bits)
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
            return np.array(list(ex.map(f, range(n))))

    def normalised(lp):  # pairs (ref allele, alt allele) per read
        r, a = lp[0::2], lp[1::2]
        out = np.empty((len(r), 2))
        for i in range(len(r)):
            out[i] = np.exp(oracle.normalize_support(float(r[i]), float(a[i])))
        return out
    base = run(0)
    nb = normalised(base)
    res = {"n_reads": n_reads, "n_pairs": n, "seed": seed, "workload": "realign_synth.generate (bench.py --workload realign)", "variants": {}}
    for name, bits in (("approx_sum3_cutoff_e-10", 1), ("stale_gap_states_of_skipped_cells", 2), ("doubled_start_mass_first_column", 4), ("all_three", 7)):
        v = run(bits)
        nv = normalised(v)
        with np.errstate(invalid="ignore"):
            d_ln = np.abs(v - base)
        res["variants"][name] = {"max_abs_dln_prob_related": float(np.nanmax(np.where(np.isfinite(d_ln), d_ln, 0.0))),
                                 "max_abs_dnormalised_support": float(np.abs(nv - nb).max()),
                                 "reads_above_1e-6": int((np.abs(nv - nb).max(axis=1) > 1e-6).sum())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
