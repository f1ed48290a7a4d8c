"""Synthetic pileup generator for the BASELINE.json configs.

Observations mimic what `varlociraptor preprocess` writes ("inputs the synthetic
generator must mimic"):
  * SNV, not realigned: prob_alt / prob_ref in {ln(1-e), ln(e) + ln 0.3333} with e = 10^(-Q/10)
    (src/variants/evidence/bases.rs:9-52, src/variants/types/snv.rs:96-116);
  * realigned (indel/MNV/SV): normalised so exp(pa)+exp(pr) = 1 or both ln 0.5
    (src/variants/evidence/realignment/mod.rs:359-374);
  * prob_missed_allele = LAE(pr, pa) - ln 2 (src/variants/types/mod.rs:100-102);
  * prob_mapping = pileup mean of {max-MAPQ prob | 0.5} (+1 pseudo observation if n < 20)
    (read_observation.rs:456-502), written adjusted (preprocessing/mod.rs:951);
  * prob_double_overlap in {0, -inf}; prob_hit_base = -ln(read_len);
  * every log-prob rounded through MiniLogProb (f16 if < -10 and same floor, else f32;
    src/utils/mod.rs:449-474).
All draws come from numpy's PCG64 seeded with 20260927 + config id (+ chunk index), so the same
(config, n_loci, seed, chunk) always yields the same batch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import abi
from .batch import PileupBatch
from .scenario import Scenario, single_sample, tumor_normal

BASE_SEED = 20260927
READ_LEN = 150
MAX_MAPQ = 60
LN_05 = np.log(0.5)


def minilogprob_round(x: np.ndarray) -> np.ndarray:
    """MiniLogProb::new then to_logprob (src/utils/mod.rs:455-473), vectorised; returns float32."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        h = x.astype(np.float16).astype(np.float64)
        use_h = (x < -10.0) & (np.floor(h) == np.floor(x))
    return np.where(use_h, h, x.astype(np.float32).astype(np.float64)).astype(np.float32)


def _lae(a, b):
    m = np.maximum(a, b)
    with np.errstate(invalid="ignore"):
        r = m + np.log1p(np.exp(-np.abs(a - b)))
    return np.where(np.isneginf(m), -np.inf, r)


@dataclass
class SynthConfig:
    name: str
    config_id: int
    scenario: Scenario
    depth: float                     # mean depth per sample
    type_mix: Dict[int, float]       # vlr_variant_type -> fraction
    classes: List[Tuple[str, float, Tuple]]  # (label, fraction, per-sample VAF spec)
    purity: Optional[float] = None
    hp_fraction: float = 0.3         # of indel loci carrying homopolymer fields
    artifact_fraction: float = 0.03  # loci with an injected systematic bias among alt reads
    other_orientation: float = 0.02
    softclip: float = 0.03
    alt_locus_fraction: float = 0.02
    empty_fraction: float = 0.0      # loci with an empty pileup in one sample (edge case)
    strand_none_fraction: float = 0.0    # observations of SV/breakend loci without strand information (Strand::None:
                                         # realignment/mod.rs:237,387-396 keeps the strand of informative reads only)
    max_depth: int = 200                 # clip of the Poisson depth; raised by the deep-pileup tests
    breakend_pair_fraction: float = 0.0  # SV loci followed by a mate record that shares their pileup (breakend groups,
                                         # calling.rs:569-580,726-741): truth["group"] gives the shared id


def config2() -> SynthConfig:
    """1 GPU, SNV loci, single-sample generic scenario, 30x (BASELINE configs[1])."""
    return SynthConfig(
        name="single-sample-snv-30x", config_id=2, scenario=single_sample(0.01), depth=30.0,
        type_mix={abi.VT_SNV: 1.0},
        classes=[("absent", 0.70, ((0.0, 0.0),)), ("subclonal", 0.20, ((0.02, 0.5),)),
                 ("het", 0.07, ((0.5, 0.5),)), ("hom", 0.03, ((1.0, 1.0),))],
    )


def config3(purity: float = 0.75, type_mix=None, config_id: int = 3) -> SynthConfig:
    """1 GPU, tumor-normal with contamination, SNV+indel, 100x (BASELINE configs[2]).
    Sample order is BTreeMap order: normal = 0, tumor = 1.  VAF spec per sample: (lo, hi) uniform."""
    return SynthConfig(
        name="tumor-normal-100x", config_id=config_id, scenario=tumor_normal(purity), depth=100.0,
        type_mix=type_mix or {abi.VT_SNV: 0.8, abi.VT_INDEL: 0.2}, purity=purity,
        classes=[
            ("absent", 0.60, ((0.0, 0.0), (0.0, 0.0))),
            ("somatic_tumor", 0.20, ((0.0, 0.0), (0.02, 0.6))),
            ("germline_het", 0.12, ((0.5, 0.5), (0.5, 0.5))),
            ("germline_hom", 0.05, ((1.0, 1.0), (1.0, 1.0))),
            ("somatic_normal", 0.03, ((0.05, 0.45), (0.05, 0.6))),
        ],
    )


def config4() -> SynthConfig:
    """8 GPU, mixed SNV/MNV/indel, tumor-normal, 100x (BASELINE configs[3]); sharded by the caller."""
    c = config3(type_mix={abi.VT_SNV: 0.70, abi.VT_MNV: 0.05, abi.VT_INDEL: 0.25}, config_id=4)
    c.name = "tumor-normal-mixed-100x"
    return c


def pedigree_scenario() -> Scenario:
    """tests/resources/prior/scenarios/pedigree.scenario.yaml on an autosome (ploidy 2 for every sample):
    ploidy-derived universes {0, 0.5, 1}, Mendelian inheritance of child and sibling from mother and father."""
    from .scenario import Inheritance, Sample, Species
    species = Species(heterozygosity=0.001, germline_mutation_rate=1e-3, ploidy=2)
    mend = Inheritance(abi.INHERIT_MENDELIAN, ("mother", "father"))
    samples = {"mother": Sample(), "father": Sample(), "child": Sample(inheritance=mend), "sibling": Sample(inheritance=mend)}
    events = {
        "denovo_child": "(child:0.5 | child:1.0) & mother:0.0 & father:0.0",
        "denovo_sibling": "(sibling:0.5 | sibling:1.0) & mother:0.0 & father:0.0",
        "inherited": "!mother:0.0 | !father:0.0",
    }
    return Scenario(samples, events, species=species)


def config5() -> SynthConfig:
    """8 GPU, 4-sample pedigree, 60x, incl. SV/breakend loci (BASELINE configs[4]).
    Sample order (BTreeMap): child 0, father 1, mother 2, sibling 3."""
    h = (0.5, 0.5)
    z = (0.0, 0.0)
    return SynthConfig(
        name="pedigree-60x", config_id=5, scenario=pedigree_scenario(), depth=60.0,
        type_mix={abi.VT_SNV: 0.65, abi.VT_INDEL: 0.25, abi.VT_SV: 0.10},
        strand_none_fraction=0.15, breakend_pair_fraction=0.5,
        classes=[("absent", 0.55, (z, z, z, z)), ("inherited_father", 0.15, (h, h, z, z)), ("inherited_mother", 0.10, (z, z, h, h)),
                 ("inherited_both", 0.08, ((1.0, 1.0), h, h, h)), ("denovo_child", 0.07, (h, z, z, z)), ("denovo_sibling", 0.05, (z, z, z, h))],
    )


CONFIGS = {"config2": config2, "config3": config3, "config4": config4, "config5": config5}


def generate(cfg: SynthConfig, n_loci: int, seed: Optional[int] = None, chunk: int = 0,
             bias_mask: int = abi.BIAS_ALL) -> PileupBatch:
    """Generate `n_loci` loci of configuration `cfg` (one chunk)."""
    seed = BASE_SEED + cfg.config_id if seed is None else seed
    rng = np.random.Generator(np.random.PCG64([seed, chunk]))
    S = len(cfg.scenario.sample_names)
    L = n_loci

    # ---- per locus
    vts = np.array(list(cfg.type_mix.keys()), np.uint8)
    vt = rng.choice(vts, size=L, p=np.array(list(cfg.type_mix.values())))
    frac = np.array([c[1] for c in cfg.classes])
    cls = rng.choice(len(cfg.classes), size=L, p=frac / frac.sum())
    vaf = np.zeros((L, S))
    for ci, (_, _, spec) in enumerate(cfg.classes):
        sel = cls == ci
        for s in range(S):
            lo, hi = spec[s]
            vaf[sel, s] = lo if lo == hi else rng.uniform(lo, hi, size=int(sel.sum()))
    is_snv = vt == abi.VT_SNV
    is_snv_or_mnv = is_snv | (vt == abi.VT_MNV)
    is_indel = vt == abi.VT_INDEL
    has_hp = is_indel & (rng.random(L) < cfg.hp_fraction)
    artifact_kind = np.where(rng.random(L) < cfg.artifact_fraction, rng.integers(1, 5, size=L), 0)  # 1 strand 2 orient 3 softclip 4 position
    has_altloc = rng.random(L) < cfg.alt_locus_fraction
    depth = np.clip(rng.poisson(cfg.depth, size=(L, S)), 1, max(200, int(getattr(cfg, "max_depth", 200))))
    if cfg.empty_fraction > 0:
        empty = rng.random((L, S)) < cfg.empty_fraction
        depth = np.where(empty, 0, depth)
    counts = depth.reshape(-1)
    obs_offset = np.zeros(L * S + 1, np.int64)
    np.cumsum(counts, out=obs_offset[1:])
    N = int(obs_offset[-1])
    pile = np.repeat(np.arange(L * S), counts)  # pileup id per obs
    loc = pile // S
    smp = pile % S

    # ---- effective alt fraction (contamination mixes the contaminant's VAF in)
    eff = vaf.copy()
    for s, name in enumerate(cfg.scenario.sample_names):
        c = cfg.scenario.samples[name].contamination
        if c is not None:
            by = cfg.scenario.idx[c.by]
            eff[:, s] = (1.0 - c.fraction) * vaf[:, s] + c.fraction * vaf[:, by]
    from_alt = rng.random(N) < eff[loc, smp]

    # ---- allele evidence
    q = rng.choice(np.array([20, 30, 37, 40]), size=N, p=[0.05, 0.15, 0.4, 0.4])
    e = 10.0 ** (-q / 10.0)
    err = rng.random(N) < e
    shows_alt = np.where(err, rng.random(N) < (1.0 / 3.0), True) & from_alt | (~from_alt & err & (rng.random(N) < (1.0 / 3.0)))
    ln_call = np.log1p(-e)
    ln_mis = np.log(e) + np.log(0.3333)
    pa_snv = np.where(shows_alt, ln_call, ln_mis)
    pr_snv = np.where(shows_alt, ln_mis, ln_call)
    # realigned: posterior-like support p for the true allele, normalised pair
    p_true = np.clip(rng.beta(12.0, 1.0, size=N), 1e-12, 1.0 - 1e-9)
    weak = rng.random(N) < 0.08  # uninformative reads: both ln 0.5
    p_alt = np.where(from_alt, p_true, 1.0 - p_true)
    pa_re = np.where(weak, LN_05, np.log(p_alt))
    pr_re = np.where(weak, LN_05, np.log1p(-p_alt))
    snv_obs = is_snv[loc]
    pa = np.where(snv_obs, pa_snv, pa_re)
    pr = np.where(snv_obs, pr_snv, pr_re)
    pa = minilogprob_round(pa).astype(np.float64)
    pr = minilogprob_round(pr).astype(np.float64)
    missed = _lae(pr, pa) - np.log(2.0)

    # ---- mapping: raw MAPQ -> pileup-mean adjustment (read_observation.rs:456-502)
    mapq = np.where(rng.random(N) < 0.95, MAX_MAPQ, rng.integers(1, MAX_MAPQ, size=N))
    # loci with alt loci tend to have low MAPQs among alt reads
    low = has_altloc[loc] & from_alt & (rng.random(N) < 0.7)
    mapq = np.where(low, rng.integers(1, 30, size=N), mapq)
    is_max = mapq == MAX_MAPQ
    max_pm = np.log1p(-(10.0 ** (-MAX_MAPQ / 10.0)))
    contrib = np.where(is_max, np.exp(max_pm), 0.5)
    sums = np.bincount(pile, weights=contrib, minlength=L * S)
    n_p = counts.astype(np.float64)
    small = counts < 20
    mean = np.where(small, (sums + 0.5) / (n_p + 1.0), sums / np.maximum(n_p, 1.0))
    with np.errstate(divide="ignore"):
        pm = np.log(mean)[pile]

    # ---- categorical features
    strand = rng.integers(0, 2, size=N).astype(np.uint32)
    double = rng.random(N) < 0.1
    pdo = np.where(double, 0.0, -np.inf)
    strand = np.where(double, abi.STRAND_BOTH, strand)
    orient = rng.integers(0, 2, size=N).astype(np.uint32)
    u = rng.random(N)
    orient = np.where(u < cfg.other_orientation, abi.ORIENT_OTHER, np.where(u < cfg.other_orientation + 0.01, abi.ORIENT_NONE, orient))
    softclip = rng.random(N) < cfg.softclip
    position = rng.integers(0, READ_LEN, size=N)
    major_pos = rng.integers(0, READ_LEN, size=L * S)
    alt_like = pa > pr
    ak = artifact_kind[loc]
    strand = np.where((ak == 1) & alt_like & ~double, abi.STRAND_FORWARD, strand)
    orient = np.where((ak == 2) & alt_like, abi.ORIENT_F1R2, orient)
    softclip = np.where((ak == 3) & alt_like, True, softclip)
    position = np.where((ak == 4) & alt_like, major_pos[pile], position)
    readpos_major = position == major_pos[pile]
    # SNV reads whose evidence is uninformative lose strand info (snv.rs:118-124): not generated here
    rng2 = np.random.Generator(np.random.PCG64([seed, chunk, 7]))  # features added later draw from their own stream
    if cfg.strand_none_fraction > 0:
        none = (vt[loc] == abi.VT_SV) & (rng2.random(N) < cfg.strand_none_fraction)
        strand = np.where(none, abi.STRAND_NONE, strand)
        pdo = np.where(none, -np.inf, pdo)
    altloc = np.full(N, abi.ALTLOCUS_NONE, np.uint32)
    al = has_altloc[loc]
    r = rng.random(N)
    altloc = np.where(al & low, np.where(r < 0.8, abi.ALTLOCUS_MAJOR, abi.ALTLOCUS_SOME), altloc)
    altloc = np.where(al & ~low & (r < 0.03), abi.ALTLOCUS_SOME, altloc)
    paired = np.ones(N, bool)
    psa = np.where(snv_obs, 0.0, np.log(rng.uniform(0.85, 1.0, size=N)))
    phb = np.full(N, -np.log(float(READ_LEN)))

    hp_obs = has_hp[loc]
    hp_len = np.where(hp_obs, rng.choice(np.array([-2, -1, 0, 0, 0, 1, 2]), size=N), -128).astype(np.int16)
    hp_len = np.where(hp_obs & from_alt & (rng.random(N) < 0.6), np.where(rng.random(N) < 0.5, 1, -1), hp_len).astype(np.int16)
    hp_art = np.where(hp_obs, np.log(rng.uniform(0.02, 1.0, size=N)), np.nan)
    hp_var = np.where(hp_obs, np.log(rng.uniform(0.02, 1.0, size=N)), np.nan)

    cols = {
        "prob_mapping": minilogprob_round(pm),
        "prob_alt": pa.astype(np.float32),
        "prob_ref": pr.astype(np.float32),
        "prob_missed_allele": minilogprob_round(missed),
        "prob_sample_alt": minilogprob_round(psa),
        "prob_double_overlap": pdo.astype(np.float32),
        "prob_hit_base": minilogprob_round(phb),
        "prob_hp_artifact": np.where(np.isnan(hp_art), np.nan, minilogprob_round(np.nan_to_num(hp_art))).astype(np.float32),
        "prob_hp_variant": np.where(np.isnan(hp_var), np.nan, minilogprob_round(np.nan_to_num(hp_var))).astype(np.float32),
        "flags": abi.pack_flags(strand, orient, readpos_major, softclip, paired, is_max, altloc, hp_len),
    }

    # ---- per-locus flags = WorkItem.check_* (calling.rs:557-567), all records precise
    lf = np.zeros(L, np.uint8)
    lf |= np.where(is_snv_or_mnv, abi.BIAS_ORIENTATION | abi.BIAS_POSITION | abi.BIAS_SOFTCLIP, 0).astype(np.uint8)
    lf |= np.uint8(abi.BIAS_STRAND | abi.BIAS_ALTLOCUS)
    lf |= np.where(has_hp, abi.BIAS_HOMOPOLYMER, 0).astype(np.uint8)
    is_sv = vt == abi.VT_SV  # imprecise SV/breakend records: only the alt-locus bias is checked (calling.rs:553-567)
    lf = np.where(is_sv, np.uint8(abi.BIAS_ALTLOCUS), lf).astype(np.uint8)
    lf &= np.uint8(bias_mask | 0xC0)
    if bias_mask & abi.BIAS_ORIENTATION:
        lf |= np.where(is_snv_or_mnv, abi.LOCUS_REMOVE_NONSTANDARD, 0).astype(np.uint8)
    lf |= np.where(is_snv, abi.LOCUS_HAS_SNV, 0).astype(np.uint8)
    bases = np.frombuffer(b"ACGT", np.uint8)
    refb = bases[rng.integers(0, 4, size=L)]
    altb = bases[(np.searchsorted(bases, refb) + rng.integers(1, 4, size=L)) % 4]
    locus = {"locus_flags": lf, "variant_type": vt.astype(np.uint8),
             "ref_base": np.where(is_snv, refb, 0).astype(np.uint8), "alt_base": np.where(is_snv, altb, 0).astype(np.uint8)}
    b = PileupBatch(S, obs_offset.astype(np.uint32), cols, locus)
    group = np.arange(L)
    if cfg.breakend_pair_fraction > 0 and L > 1:
        # the mate record of a breakend shares the pileup of its partner: locus k + 1 becomes a copy of locus k
        first = np.nonzero((vt[:-1] == abi.VT_SV) & (rng2.random(L - 1) < cfg.breakend_pair_fraction))[0]
        first = first[np.concatenate(([True], np.diff(first) > 1))] if len(first) else first  # no chains
        src = np.arange(L)
        src[first + 1] = first
        b = b.select(src)
        cls, vaf, group = cls[src], vaf[src], src
    b.truth = {"class": cls, "vaf": vaf, "class_names": [c[0] for c in cfg.classes], "group": group}
    return b
