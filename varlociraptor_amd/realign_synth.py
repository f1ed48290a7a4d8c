"""Synthetic (allele window, read window) pairs for the realignment kernel: a random reference window, a variant
(SNV / MNV / insertion / deletion), reads drawn from the reference or the alt allele with sequencing errors at the rate
their base qualities state, windows as Realigner::candidate_region cuts them (mod.rs:57-147: read window of at most 2 x
realignment_window = 128 bases, reference window of 1.5 x realignment_window on both sides of the breakpoint)."""
import numpy as np

from . import realign

BASES = np.frombuffer(b"ACGT", np.uint8)


def make_locus(rng, window=64, kind=None):
    ref_window = int(window * 1.5)
    ref = BASES[rng.integers(0, 4, 4 * ref_window + 200)].tobytes()
    bp = len(ref) // 2
    ref_offset, ref_end = bp - ref_window, bp + ref_window
    kind = kind or rng.choice(["snv", "mnv", "ins", "del"], p=[0.2, 0.1, 0.35, 0.35])
    if kind == "snv":
        alt = int(BASES[(np.searchsorted(BASES, ref[bp]) + 1 + rng.integers(3)) % 4])
        allele = realign.snv_allele(ref, ref_offset, ref_end, bp, alt)
        full_alt = ref[:bp] + bytes([alt]) + ref[bp + 1:]
    elif kind == "mnv":
        n = int(rng.integers(2, 5))
        alt = bytes(int(BASES[(np.searchsorted(BASES, ref[bp + k]) + 1 + rng.integers(3)) % 4]) for k in range(n))
        allele = realign.mnv_allele(ref, ref_offset, ref_end, bp, alt)
        full_alt = ref[:bp] + alt + ref[bp + n:]
    elif kind == "ins":
        n = int(rng.integers(1, 25))
        ins = BASES[rng.integers(0, 4, n)].tobytes()
        allele = realign.insertion_allele(ref, ref_offset, ref_end, bp, ins)
        full_alt = ref[:bp + 1] + ins + ref[bp + 1:]
    else:
        n = int(rng.integers(1, 40))
        allele = realign.deletion_allele(ref, ref_offset, ref_end, bp, n)
        full_alt = ref[:bp + 1] + ref[bp + 1 + n:]
    return {"kind": kind, "ref": ref, "bp": bp, "ref_allele": realign.ref_allele(ref, ref_offset, ref_end), "alt_allele": allele, "full_alt": full_alt}


def make_read(rng, locus, from_alt, window=64, quals=(20, 30, 37, 40)):
    src = locus["full_alt"] if from_alt else locus["ref"]
    n = int(rng.integers(window, 2 * window + 1))
    n = min(n, realign.MAX_PATTERN_LEN)
    start = locus["bp"] - int(rng.integers(n // 4, 3 * n // 4))
    seq = bytearray(src[start:start + n])
    q = rng.choice(quals, len(seq), p=[0.05, 0.15, 0.4, 0.4]).astype(np.uint8)
    err = rng.random(len(seq)) < 10.0 ** (-q.astype(np.float64) / 10.0)
    for k in np.nonzero(err)[0]:
        seq[k] = int(BASES[(np.searchsorted(BASES, seq[k]) + 1 + rng.integers(3)) % 4])
    return bytes(seq), q.tobytes()


def generate(n_reads, seed=1, window=64, reads_per_locus=50, banded=True):
    """PairBatch of 2 pairs per read (reference allele, alt allele), bands from the edit-distance pre-filter."""
    rng = np.random.default_rng(seed)
    pb = realign.PairBatch()
    truth = []
    locus = None
    for k in range(n_reads):
        if k % reads_per_locus == 0:
            locus = make_locus(rng, window)
        from_alt = bool(rng.random() < 0.4)
        seq, q = make_read(rng, locus, from_alt, window)
        for allele in (locus["ref_allele"], locus["alt_allele"]):
            band = -1
            if banded:
                hit = realign.best_hit(seq, allele)
                band = hit[0] + realign.EDIT_BAND
            pb.add(allele, seq, q, band)
        truth.append(from_alt)
    return pb, np.array(truth)
