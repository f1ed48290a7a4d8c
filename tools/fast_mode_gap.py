"""How far is the `fast` realignment mode's upper bound from what a concrete traceback would give?  

vlr_realign_fast_batch returns the best path probability over ALL alignments of minimal semiglobal edit distance; the reference
(PathHMMRealigner, realignment/mod.rs:547-678) evaluates the alignments bio's Myers traceback hands it, whose choice among
co-optimal alignments is unspecified.  On the bench workload's pairs this tool counts the pairs with more than one co-optimal
alignment and measures the gap between the upper bound and a fixed diagonal-first traceback — in ln P of a pair and in the
NORMALISED ref/alt supports the observation records carry.  CPU only (oracle).   usage: python tools/fast_mode_gap.py [reads]"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from varlociraptor_amd import realign_synth
from varlociraptor_amd.realign import GapParams, normalize_support

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
oracle.build()
pb, truth = realign_synth.generate(n_reads, seed=7)
gap = GapParams()
g = [gap.prob_insertion_artifact, gap.prob_deletion_artifact, gap.prob_insertion_extend_artifact, gap.prob_deletion_extend_artifact]
best = np.empty(len(pb)); fixed = np.empty(len(pb)); nco = np.empty(len(pb))
for k in range(len(pb)):
    best[k] = oracle.pathhmm_best(pb.x[k], pb.y[k], pb.q[k], g)
    fixed[k], nco[k] = oracle.pathhmm_fixed_traceback(pb.x[k], pb.y[k], pb.q[k], g)
gap_ln = best - fixed
assert (gap_ln > -1e-9).all(), "the maximum over all optimal alignments cannot be below one of them"
sup = []
for k in range(len(truth)):
    b = normalize_support(best[2 * k], best[2 * k + 1]); f = normalize_support(fixed[2 * k], fixed[2 * k + 1])
    sup.append(max(abs(math.exp(b[0]) - math.exp(f[0])), abs(math.exp(b[1]) - math.exp(f[1]))))
sup = np.array(sup)
out = {"pairs": int(len(pb)), "reads": int(len(truth)),
       "frac_pairs_with_cooptimal_alignments": float((nco > 1).mean()), "median_cooptimal_alignments_when_any": float(np.median(nco[nco > 1])) if (nco > 1).any() else 1.0,
       "frac_pairs_where_the_bound_is_not_attained_by_the_fixed_traceback": float((gap_ln > 1e-12).mean()),
       "max_gap_ln_p": float(gap_ln.max()), "mean_gap_ln_p_when_positive": float(gap_ln[gap_ln > 1e-12].mean()) if (gap_ln > 1e-12).any() else 0.0,
       "max_abs_d_normalised_support": float(sup.max()), "frac_reads_support_moves_above_1e-6": float((sup > 1e-6).mean()),
       "rule": "first end position of minimal distance, traceback prefers diagonal, then deletion, then insertion"}
print(json.dumps(out, indent=1))
