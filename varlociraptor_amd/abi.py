"""ctypes mirror of include/vlr.h (the C ABI of the engine).

Only struct layouts and constants live here; nothing is computed.  Both the product binding
(`varlociraptor_amd.engine`) and the test-only oracle binding (`oracle/oracle.py`) use these
definitions, because the oracle consumes exactly the same boundary structs as the engine.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 6
MAX_SAMPLES = 16
N_BIAS = 6

# status codes
OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_INVALID_PRIOR, ERR_OOM = 0, -1, -2, -3, -4, -5, -6
LOCUS_NAN, LOCUS_UNDERFLOW, LOCUS_TABLE_FULL, LOCUS_TOO_DEEP = 1, 2, 4, 8
LOCUS_MISSING_DATA, LOCUS_SINGLETON_ADJ, LOCUS_FILTERED_ALN = 16, 32, 64

SPECTRUM_SET, SPECTRUM_RANGE = 0, 1
CMP_EQUAL, CMP_GREATER, CMP_GREATER_EQUAL, CMP_LESS, CMP_LESS_EQUAL, CMP_NOT_EQUAL = range(6)
NODE_SAMPLE, NODE_LFC, NODE_VARIANT, NODE_TRUE, NODE_FALSE = range(5)
INHERIT_NONE, INHERIT_MENDELIAN, INHERIT_CLONAL, INHERIT_SUBCLONAL = range(4)
VT_SNV, VT_MNV, VT_INDEL, VT_SV, VT_OTHER = range(5)

# packed observation flags
F_STRAND_SHIFT, F_ORIENT_SHIFT = 0, 2
F_READPOS_MAJOR, F_SOFTCLIPPED, F_PAIRED, F_MAX_MAPQ = 1 << 4, 1 << 5, 1 << 6, 1 << 7
F_ALTLOCUS_SHIFT = 8
F_HP_LEN_VALID = 1 << 10
F_HP_LEN_SHIFT = 16
STRAND_FORWARD, STRAND_REVERSE, STRAND_BOTH, STRAND_NONE = range(4)
ORIENT_F1R2, ORIENT_F2R1, ORIENT_NONE, ORIENT_OTHER = range(4)
ALTLOCUS_MAJOR, ALTLOCUS_SOME, ALTLOCUS_NONE = range(3)

BIAS_STRAND, BIAS_ORIENTATION, BIAS_POSITION, BIAS_SOFTCLIP, BIAS_HOMOPOLYMER, BIAS_ALTLOCUS = (1 << i for i in range(6))
BIAS_ALL = 0x3F
LOCUS_REMOVE_NONSTANDARD = 1 << 6
LOCUS_HAS_SNV = 1 << 7


class Spectrum(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("set_offset", C.c_int32), ("set_len", C.c_int32),
        ("left_exclusive", C.c_int32), ("right_exclusive", C.c_int32), ("_pad", C.c_int32),
        ("start", C.c_double), ("end", C.c_double),
    ]


class Node(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("sample", C.c_int32), ("sample_b", C.c_int32), ("cmp", C.c_int32),
        ("lfc_value", C.c_double), ("vafs", Spectrum), ("positive", C.c_int32),
        ("refbase", C.c_uint8), ("altbase", C.c_uint8), ("_pad", C.c_uint8 * 2),
        ("child_offset", C.c_int32), ("n_children", C.c_int32),
    ]


class Inheritance(C.Structure):
    _fields_ = [("kind", C.c_int32), ("from0", C.c_int32), ("from1", C.c_int32), ("somatic", C.c_int32)]


class ScenarioDesc(C.Structure):
    _fields_ = [
        ("n_samples", C.c_int32),
        ("resolution", C.POINTER(C.c_double)),
        ("contaminated_by", C.POINTER(C.c_int32)),
        ("contamination_fraction", C.POINTER(C.c_double)),
        ("universe_offset", C.POINTER(C.c_int32)),
        ("universe", C.POINTER(Spectrum)),
        ("uniform_prior", C.POINTER(C.c_uint8)),
        ("ploidy", C.POINTER(C.c_int32)),
        ("germline_mutation_rate", C.POINTER(C.c_double)),
        ("somatic_effective_mutation_rate", C.POINTER(C.c_double)),
        ("inheritance", C.POINTER(Inheritance)),
        ("heterozygosity", C.c_double),
        ("fraction_indel", C.c_double), ("fraction_mnv", C.c_double), ("fraction_sv", C.c_double),
        ("is_absent_only", C.c_int32),
        ("n_events", C.c_int32),
        ("event_names", C.POINTER(C.c_char_p)),
        ("event_root_offset", C.POINTER(C.c_int32)),
        ("root_index", C.POINTER(C.c_int32)),
        ("n_nodes", C.c_int32),
        ("nodes", C.POINTER(Node)),
        ("child_index", C.POINTER(C.c_int32)),
        ("vafs", C.POINTER(C.c_double)),
        ("variant_heterozygosity_ln", C.c_double),
        ("variant_somatic_effective_mutation_rate_ln", C.c_double),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int64), ("n_samples", C.c_int32), ("_pad", C.c_int32), ("n_obs", C.c_int64),
        ("obs_offset", C.c_void_p),
        ("prob_mapping", C.c_void_p), ("prob_alt", C.c_void_p), ("prob_ref", C.c_void_p),
        ("prob_missed_allele", C.c_void_p), ("prob_sample_alt", C.c_void_p),
        ("prob_double_overlap", C.c_void_p), ("prob_hit_base", C.c_void_p),
        ("prob_hp_artifact", C.c_void_p), ("prob_hp_variant", C.c_void_p),
        ("flags", C.c_void_p),
        ("locus_flags", C.c_void_p), ("variant_type", C.c_void_p), ("ref_base", C.c_void_p), ("alt_base", C.c_void_p),
    ]


class Results(C.Structure):
    _fields_ = [
        ("n_loci", C.c_int64), ("n_out", C.c_int32), ("n_samples", C.c_int32),
        ("ln_posterior", C.c_void_p), ("ln_marginal", C.c_void_p), ("map_vaf", C.c_void_p),
        ("map_bias", C.c_void_p), ("best_event", C.c_void_p), ("status", C.c_void_p),
        ("afd_capacity", C.c_int32), ("_pad", C.c_int32),
        ("afd_count", C.c_void_p), ("afd_vaf", C.c_void_p), ("afd_lnprob", C.c_void_p),
        ("afd_text", C.c_void_p), ("afd_text_capacity", C.c_uint64), ("afd_text_span", C.c_void_p),
    ]


# column name -> dtype of the SoA observation columns of vlr_batch
OBS_COLUMNS = [
    ("prob_mapping", np.float32), ("prob_alt", np.float32), ("prob_ref", np.float32),
    ("prob_missed_allele", np.float32), ("prob_sample_alt", np.float32),
    ("prob_double_overlap", np.float32), ("prob_hit_base", np.float32),
    ("prob_hp_artifact", np.float32), ("prob_hp_variant", np.float32), ("flags", np.uint32),
]
LOCUS_COLUMNS = [("locus_flags", np.uint8), ("variant_type", np.uint8), ("ref_base", np.uint8), ("alt_base", np.uint8)]


def pack_flags(strand, orientation, readpos_major, softclipped, paired, max_mapq, alt_locus, hp_len=None):
    """Pack per-observation categorical features into the VLR_F_* bit layout (numpy arrays in, uint32 out)."""
    f = (np.asarray(strand, np.uint32) << F_STRAND_SHIFT) | (np.asarray(orientation, np.uint32) << F_ORIENT_SHIFT)
    f = f | np.where(readpos_major, F_READPOS_MAJOR, 0).astype(np.uint32)
    f = f | np.where(softclipped, F_SOFTCLIPPED, 0).astype(np.uint32)
    f = f | np.where(paired, F_PAIRED, 0).astype(np.uint32)
    f = f | np.where(max_mapq, F_MAX_MAPQ, 0).astype(np.uint32)
    f = f | (np.asarray(alt_locus, np.uint32) << F_ALTLOCUS_SHIFT)
    if hp_len is not None:
        hp = np.asarray(hp_len)
        valid = hp > -128  # -128 encodes None
        f = f | np.where(valid, F_HP_LEN_VALID, 0).astype(np.uint32)
        f = f | np.where(valid, (hp.astype(np.int8).view(np.uint8).astype(np.uint32)) << F_HP_LEN_SHIFT, 0).astype(np.uint32)
    return f.astype(np.uint32)
