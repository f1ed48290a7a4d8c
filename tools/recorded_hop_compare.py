"""What the observations RECORDED in the reference's test_nanopore_05 testcase say about the homopolymer-mode pair HMM (run here, where
/root/reference exists; round 6).  The testcase is single-end (26 long reads), so its candidates.vcf holds one (PROB_ALT, PROB_REF)
pair per READ — the only recorded values of the collection that can be compared read by read with a realignment of the BAM (the other
testcases with a BAM are paired-end: one value per fragment, with the insert-size term).  They were written in observation format 13
(the tree reads format 15) by the version of the reporter's site; the realignment is recomputed by the reference's test.

Prints, per read, the normalised supports of oracle/vlr_realign_oracle.cpp (vlro_homopoly_prob_related, gap and homopolymer-run
parameters of the testcase's alignment properties) next to the recorded ones.  Result (profiles/r06g_experiments.md section 5): the file
refutes a hop state that emits the repeated read base like an inserted base (P(miscall) = 10^-25.5 at quality 255: 58 nats where the
file has one hop probability) — the restatement and the kernel emit it like a matching base since — and it cannot pin the rest: it
was written by another version (other roles of the seq / ref run probabilities, other supports for reads without a run-length change).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bam_pairs as bp
from varlociraptor_amd import obsfmt
from varlociraptor_amd.realign import HopParams

lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "libvlr_oracle.so"))
lib.vlro_homopoly_prob_related.restype = C.c_double
lib.vlro_homopoly_prob_related.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]


def hp(x, y, q, g, h):
    xa, ya, qa = np.frombuffer(x, np.uint8), np.frombuffer(y, np.uint8), np.asarray(bytearray(q), np.uint8)
    return float(lib.vlro_homopoly_prob_related(xa.ctypes.data, len(x), ya.ctypes.data, qa.ctypes.data, len(y), (C.c_double * 4)(*g), (C.c_double * 16)(*h), -1))


rec = [l for l in open("/root/reference/tests/resources/testcases/test_nanopore_05/candidates.vcf") if not l.startswith("#")][0].rstrip("\n").split("\t")
info = dict(kv.split("=", 1) for kv in rec[7].split(";") if "=" in kv)
ra = obsfmt._vec_minilogprob(obsfmt._bytes_from_info(info["PROB_ALT"]))
rr = obsfmt._vec_minilogprob(obsfmt._bytes_from_info(info["PROB_REF"]))
case = bp.indel_pairs(os.path.join(ROOT, "tests", "golden", "bam", "test_nanopore_05"))
spec = bp.BAM_CASES["test_nanopore_05"]
g, h = list(spec["gap"]), HopParams(*spec["hop"]).as_list()
worst = 0.0
for k, (r, seq, q, refal) in enumerate(case.reads):
    lr, la = hp(refal, seq, bytes(q), g, h), hp(case.alt_allele, seq, bytes(q), g, h)
    t = np.logaddexp(lr, la)
    worst = max(worst, abs(la - t - ra[k]), abs(lr - t - rr[k]))
    print("%2d %-24s %-18s restated pa %7.3f pr %7.3f | recorded pa %7.3f pr %7.3f" % (k, r.qname, "".join("%d%s" % (l, o) for o, l in r.cigar), la - t, lr - t, ra[k], rr[k]))
print("largest |restated - recorded|: %.2f" % worst)
