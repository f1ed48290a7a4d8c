/*
 * vlr.h — C ABI of the MI355X-native per-locus Bayesian likelihood engine
 * (drop-in for the model-evaluation path of `varlociraptor call variants`).
 *
 * The reference (Rust, /root/reference = varlociraptor v8.9.3) has no FFI boundary;
 * the path sits behind the in-process trait boundary
 *     bio::stats::bayesian::model::Model::compute(events, &Data) -> ModelInstance
 * invoked once per record at src/calling/variants/calling.rs:760 (inside
 * Caller::call_record, calling.rs:720-842) and consumed by calling.rs:762-813 and
 * sample_infos (calling.rs:844-937).  This header is what a Rust `extern "C"` block
 * in calling.rs would bind instead: one *plan* per
 * (scenario, contig) replaces configure_model (calling.rs:632-718) and one
 * *batch run* replaces the per-record call_record loop for many records at once.
 *
 * Conventions
 *   - all functions return 0 (VLR_OK) or a negative vlr_status error code;
 *     vlr_last_error() gives a thread-local human readable message.
 *   - numeric invariants the reference enforces with assert!/panic!
 *     (likelihood.rs:113,150,191,217; modes/generic.rs:228) are surfaced per locus
 *     in vlr_results.status instead of aborting.
 *   - no torch / C++ types; plain pointers and sizes only.  Pointers inside
 *     vlr_batch / vlr_results are DEVICE pointers for vlr_batch_run and HOST
 *     pointers for vlr_batch_run_host.
 *   - results are in input order; loci are independent.
 *   - thread-safe across plans; a plan may be used from one thread at a time.
 */
#ifndef VLR_H
#define VLR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLR_ABI_VERSION 6   /* 3: vlr_plan_reserve takes the AFD capacity, 30 named events, vlr_node_*, homopolymer realignment; 4: device front door;
                             * 5: sharded device reader, calls-file parts, vlr_ingest_device_trim, CRC32 of BGZF members checked by both readers;
                             * 6: calls emission on the device — vlr_results.afd_text (FORMAT/AFD text), OBS text in the observation summaries,
                             *    vlr_obs_table_summaries */
#define VLR_MAX_SAMPLES 16     /* samples per scenario supported by the device path   */
#define VLR_N_BIAS      6      /* strand, orientation, position, softclip, homopolymer, alt-locus */

/* ---------------------------------------------------------------- status codes */
typedef enum {
    VLR_OK                    = 0,
    VLR_ERR_INVALID_ARGUMENT  = -1,
    VLR_ERR_UNSUPPORTED       = -2,  /* scenario feature outside the device path */
    VLR_ERR_NO_DEVICE         = -3,  /* HIP runtime/device missing: the product path never falls back to CPU */
    VLR_ERR_HIP               = -4,
    VLR_ERR_INVALID_PRIOR     = -5,  /* prior.rs:788-825 CheckablePrior::check */
    VLR_ERR_OUT_OF_MEMORY     = -6
} vlr_status;

/* per-locus status bits (vlr_results.status) */
#define VLR_LOCUS_OK            0u
#define VLR_LOCUS_NAN           (1u << 0)  /* a density/likelihood became NaN (reference: panic)        */
#define VLR_LOCUS_UNDERFLOW     (1u << 1)  /* a per-observation likelihood left the f64 linear range    */
#define VLR_LOCUS_TABLE_FULL    (1u << 2)  /* visited-point table of a range chain overflowed           */
#define VLR_LOCUS_TOO_DEEP      (1u << 3)  /* pileup larger than the plan's LDS budget                  */
#define VLR_LOCUS_MISSING_DATA  (1u << 4)  /* all pileups empty (calling/variants/mod.rs:423-430)       */
#define VLR_LOCUS_SINGLETON_ADJ (1u << 5)  /* Hint::AdjustedSingletonEvidence (calling.rs:613-617)      */
#define VLR_LOCUS_FILTERED_ALN  (1u << 6)  /* Hint::FilteredNonStandardAlignments (calling.rs:618-622)  */

/* ---------------------------------------------------------------- scenario description
 * Flattened form of grammar::Scenario after normalisation for ONE contig
 * (grammar/mod.rs:129-279, grammar/vaftree.rs:168-305).  Samples are indexed in
 * BTreeMap key order (grammar/mod.rs:178-190).                                         */

typedef enum { VLR_SPECTRUM_SET = 0, VLR_SPECTRUM_RANGE = 1 } vlr_spectrum_kind;

/* grammar/formula.rs:1018-1047 VAFSpectrum, 1049-1056 VAFRange */
typedef struct {
    int32_t kind;            /* vlr_spectrum_kind */
    int32_t set_offset;      /* SET: first member in vlr_scenario_desc.vafs (ascending)  */
    int32_t set_len;         /* SET: number of members                                  */
    int32_t left_exclusive;  /* RANGE */
    int32_t right_exclusive; /* RANGE */
    int32_t _pad;
    double  start, end;      /* RANGE */
} vlr_spectrum;

/* utils/comparison.rs:4-12 */
typedef enum {
    VLR_CMP_EQUAL = 0, VLR_CMP_GREATER = 1, VLR_CMP_GREATER_EQUAL = 2,
    VLR_CMP_LESS = 3, VLR_CMP_LESS_EQUAL = 4, VLR_CMP_NOT_EQUAL = 5
} vlr_cmp;

/* grammar/vaftree.rs:63-81 NodeKind */
typedef enum {
    VLR_NODE_SAMPLE = 0, VLR_NODE_LFC = 1, VLR_NODE_VARIANT = 2,
    VLR_NODE_TRUE = 3, VLR_NODE_FALSE = 4
} vlr_node_kind;

typedef struct {
    int32_t kind;          /* vlr_node_kind                                              */
    int32_t sample;        /* SAMPLE: sample index; LFC: sample_a                        */
    int32_t sample_b;      /* LFC                                                        */
    int32_t cmp;           /* LFC: vlr_cmp (utils/log2_fold_change.rs:28-31)             */
    double  lfc_value;     /* LFC                                                        */
    vlr_spectrum vafs;     /* SAMPLE                                                     */
    int32_t positive;      /* VARIANT                                                    */
    uint8_t refbase;       /* VARIANT: IUPAC code (grammar/formula.rs:20-43)             */
    uint8_t altbase;
    uint8_t _pad[2];
    int32_t child_offset;  /* children = child_index[child_offset .. +n_children]        */
    int32_t n_children;
} vlr_node;

/* variants/model/prior.rs:41-46 */
typedef enum {
    VLR_INHERIT_NONE = 0, VLR_INHERIT_MENDELIAN = 1, VLR_INHERIT_CLONAL = 2, VLR_INHERIT_SUBCLONAL = 3
} vlr_inheritance_kind;

typedef struct {
    int32_t kind;      /* vlr_inheritance_kind */
    int32_t from0;     /* parent sample index  */
    int32_t from1;     /* second parent (mendelian) */
    int32_t somatic;   /* clonal: somatic flag */
} vlr_inheritance;

/* variants/model/mod.rs:139-160 VariantType, collapsed to what the path distinguishes
 * (grammar/mod.rs:420-431 VariantTypeFraction::get, calling.rs:517-534 is_snv_or_mnv) */
typedef enum {
    VLR_VT_SNV = 0, VLR_VT_MNV = 1, VLR_VT_INDEL = 2 /* INS|DEL|REP */, VLR_VT_SV = 3 /* INV|BND|DUP */,
    VLR_VT_OTHER = 4 /* METH, REF */, VLR_N_VARIANT_TYPES = 5
} vlr_variant_type;

typedef struct {
    int32_t n_samples;
    /* per sample [n_samples] */
    const double*  resolution;             /* grammar/mod.rs:445-447 (default 0.01)                    */
    const int32_t* contaminated_by;        /* -1: SampleModel::Normal; else Contaminated{by} (modes/generic.rs:464-470) */
    const double*  contamination_fraction; /* purity = 1 - fraction (modes/generic.rs:482-484)         */
    const int32_t* universe_offset;        /* [n_samples+1] into universe[]; Sample::contig_universe (grammar/mod.rs:503-579) */
    const vlr_spectrum* universe;
    /* prior (variants/model/prior.rs:61-80; calling.rs:1072-1087) */
    const uint8_t* uniform_prior;          /* sample declares `universe`                               */
    const int32_t* ploidy;                 /* -1 = None                                                */
    const double*  germline_mutation_rate;            /* NaN = None */
    const double*  somatic_effective_mutation_rate;   /* NaN = None */
    const vlr_inheritance* inheritance;
    double  heterozygosity;                /* species heterozygosity as plain probability; NaN = None  */
    double  fraction_indel, fraction_mnv, fraction_sv;  /* VariantTypeFraction (defaults 0.0125/0.001/0.01) */
    int32_t is_absent_only;                /* !--full-prior (calling.rs:1086)                          */
    /* event universe of the scenario, WITHOUT `absent` and without artifact twins
     * (both are added by the engine exactly like calling.rs:654-687); events must be given
     * in ascending name order (BTreeMap order of grammar/mod.rs:137).  At most 62 (more than 30: the
     * wide build of the kernels; VLR_ERR_UNSUPPORTED above).                                          */
    int32_t n_events;
    const char* const* event_names;
    const int32_t* event_root_offset;      /* [n_events+1] into root_index[]                           */
    const int32_t* root_index;             /* node ids of the VAFTree roots of each event              */
    int32_t n_nodes;
    const vlr_node* nodes;
    const int32_t* child_index;
    const double*  vafs;                   /* pool for SET spectra                                     */
    /* per-variant prior overrides from the candidate record's INFO HETEROZYGOSITY / SOMATIC_EFFECTIVE_MUTATION_RATE
     * (PHRED in the file, calling.rs:470-494), as LogProb; NaN = None.  The reference installs them together with the
     * contig's universe (calling.rs:704-713), i.e. they are plan data: when present they replace the (variant-type
     * scaled) species heterozygosity / the per-sample somatic rates (prior.rs:250-270).                          */
    double  variant_heterozygosity_ln;
    double  variant_somatic_effective_mutation_rate_ln;
} vlr_scenario_desc;

/* ---------------------------------------------------------------- observations (SoA)
 * One entry per ReadObservation (variants/evidence/observations/read_observation.rs:219-278)
 * as decoded from observation-format v15 (calling/variants/preprocessing/mod.rs:818-919).
 * All log-probabilities are MiniLogProb f16/f32 in the wire format (utils/mod.rs:449-474), so
 * f32 columns are lossless.  prob_mismapping / prob_single_overlap are derived
 * (read_observation.rs:283-292).                                                                     */

/* packed per-observation flags */
#define VLR_F_STRAND_SHIFT   0   /* 2 bits: 0 Forward 1 Reverse 2 Both 3 None (read_observation.rs:51-57)         */
#define VLR_F_ORIENT_SHIFT   2   /* 2 bits: 0 F1R2 1 F2R1 2 None 3 other/non-standard (bio_types SequenceReadPairOrientation) */
#define VLR_F_READPOS_MAJOR  (1u << 4)   /* ReadPosition::Major (read_observation.rs:125-129)                     */
#define VLR_F_SOFTCLIPPED    (1u << 5)
#define VLR_F_PAIRED         (1u << 6)
#define VLR_F_MAX_MAPQ       (1u << 7)
#define VLR_F_ALTLOCUS_SHIFT 8   /* 2 bits: 0 Major 1 Some 2 None (read_observation.rs:213-217)                   */
#define VLR_F_HP_LEN_VALID   (1u << 10)  /* homopolymer_indel_len is Some                                         */
#define VLR_F_HP_LEN_SHIFT   16  /* 8 bits: homopolymer_indel_len as int8                                         */

#define VLR_STRAND_FORWARD 0
#define VLR_STRAND_REVERSE 1
#define VLR_STRAND_BOTH    2
#define VLR_STRAND_NONE    3
#define VLR_ORIENT_F1R2    0
#define VLR_ORIENT_F2R1    1
#define VLR_ORIENT_NONE    2
#define VLR_ORIENT_OTHER   3
#define VLR_ALTLOCUS_MAJOR 0
#define VLR_ALTLOCUS_SOME  1
#define VLR_ALTLOCUS_NONE  2

/* bias-enable mask per locus = WorkItem.check_* (calling.rs:557-567) */
#define VLR_BIAS_STRAND      (1u << 0)
#define VLR_BIAS_ORIENTATION (1u << 1)
#define VLR_BIAS_POSITION    (1u << 2)
#define VLR_BIAS_SOFTCLIP    (1u << 3)
#define VLR_BIAS_HOMOPOLYMER (1u << 4)
#define VLR_BIAS_ALTLOCUS    (1u << 5)
/* locus flag: Pileup::remove_nonstandard_alignments applies (is_snv_or_mnv && !omit_read_orientation_bias;
 * calling.rs:590-598, pileup.rs:26-43) */
#define VLR_LOCUS_REMOVE_NONSTANDARD (1u << 6)
/* locus flag: record is an SNV (ref/alt single base) so Data.snv is Some (calling.rs:517-524) */
#define VLR_LOCUS_HAS_SNV    (1u << 7)

typedef struct {
    int64_t n_loci;
    int32_t n_samples;
    int32_t _pad;
    int64_t n_obs;                    /* total observations = obs_offset[n_loci*n_samples]               */
    /* pileup p = locus*n_samples + sample covers observations [obs_offset[p], obs_offset[p+1])           */
    const uint32_t* obs_offset;       /* [n_loci*n_samples + 1]                                           */
    const float* prob_mapping;        /* ln P(correctly mapped), already MAPQ-adjusted (preprocessing/mod.rs:951) */
    const float* prob_alt;
    const float* prob_ref;
    const float* prob_missed_allele;
    const float* prob_sample_alt;
    const float* prob_double_overlap;
    const float* prob_hit_base;
    const float* prob_hp_artifact;    /* prob_observable_at_homopolymer_artifact; NaN = None; column may be NULL */
    const float* prob_hp_variant;     /* prob_observable_at_homopolymer_variant;  NaN = None; column may be NULL */
    const uint32_t* flags;            /* VLR_F_*                                                          */
    /* per locus */
    const uint8_t* locus_flags;       /* VLR_BIAS_* | VLR_LOCUS_REMOVE_NONSTANDARD | VLR_LOCUS_HAS_SNV     */
    const uint8_t* variant_type;      /* vlr_variant_type (prior's variant_type_fraction)                 */
    const uint8_t* ref_base;          /* SNV only (modes/generic.rs:18-22)                                */
    const uint8_t* alt_base;
} vlr_batch;

/* ---------------------------------------------------------------- results
 * What calling.rs:762-813 + sample_infos (844-937) extract from the ModelInstance.                     */
typedef struct {
    int64_t n_loci;
    int32_t n_out;          /* = n_events + 2                                                             */
    int32_t n_samples;
    /* ln posterior probabilities [n_loci * n_out]: column 0 = `absent`, 1..n_events = scenario events
     * (clean twins, in the order of vlr_scenario_desc), n_events+1 = `artifact`
     * (ln-sum over artifact twins; calling.rs:785-799).  PROB_* in the BCF = -10/ln10 * value.           */
    double*  ln_posterior;
    double*  ln_marginal;   /* [n_loci] ln marginal of Model::compute (may be NULL)                       */
    double*  map_vaf;       /* [n_loci * n_samples] FORMAT/AF: MAP allele frequency; 0 if MAP is an artifact (calling.rs:875-887); NaN if no MAP */
    /* [n_loci * VLR_N_BIAS] bias state of the MAP estimate, identical for all samples of a locus
     * (modes/generic.rs:248-257): 0 = none, else strand 1 '+',2 '-'; orientation 1 '>',2 '<'; others 1.  */
    uint8_t* map_bias;
    int32_t* best_event;    /* [n_loci] index into the engine's event universe: 0 absent, 1+2*e clean e, 2+2*e artifact twin (may be NULL) */
    uint32_t* status;       /* [n_loci] VLR_LOCUS_* bits                                                   */
    /* optional AFD (calling.rs:889-928; FORMAT/AFD): for locus l and sample s the entries
     * afd_vaf/afd_lnprob[(l*S+s)*afd_capacity .. + min(afd_count[l*S+s], afd_capacity)]; unordered;
     * all NULL to skip.                                                                                  */
    int32_t  afd_capacity;
    int32_t  _pad;
    int32_t* afd_count;     /* [n_loci * n_samples]                                                        */
    double*  afd_vaf;       /* [n_loci * n_samples * afd_capacity] (f64: the BCF prints it with 3 decimals)      */
    double*  afd_lnprob;    /* [n_loci * n_samples * afd_capacity]                                         */
    /* (ABI 6) optional FORMAT/AFD TEXT, written on the device by vlr_batch_run_host / vlr_batch_run_device_in / vlr_node_batch_run_host
     * (HOST pointers; needs the afd_* buffers above): the list of locus l and sample s in the order and format of
     * Call::write_final_record (calling/variants/mod.rs:473-559) — entries by ascending allele frequency (stable), "%.3f=%.2f" of
     * (vaf, -10 ln p / ln 10) joined by ',' — at afd_text[afd_text_span[2 p] .. + afd_text_span[2 p + 1]], p = l * n_samples + s.
     * A length of 0xffffffff: not formatted (a value beyond 1e12, a NaN allele frequency, more than 1 024 entries, text buffer full) —
     * the list is in afd_vaf / afd_lnprob then, which are only copied to the host when some list of the call needs them.
     * vlr_calls_write / vlr_calls_writer_append take the text where it is given.  NULL: no text. */
    uint8_t*  afd_text;
    uint64_t  afd_text_capacity;
    uint32_t* afd_text_span;   /* [n_loci * n_samples * 2] */
} vlr_results;

/* ---------------------------------------------------------------- entry points */
typedef struct vlr_plan vlr_plan;

/* ABI version of the loaded library. */
int  vlr_abi_version(void);
/* Identity of the build: "<sha1 of the engine sources and the Makefile>" (measurement files under profiles/ carry the id of
 * the build they were taken from; the build-matrix test refuses variants built from other sources). */
const char* vlr_build_id(void);
/* Thread-local message of the last error. */
const char* vlr_last_error(void);

/* Compile a scenario (one contig) into a device plan: flattened VAFTrees, event universe incl.
 * `absent` and artifact twins, prior table, contamination wiring.  Replaces
 * Caller::configure_model (calling.rs:632-718) + GenericModelBuilder::build (modes/generic.rs:93-105). */
int  vlr_plan_create(const vlr_scenario_desc* scenario, int device, vlr_plan** out);
void vlr_plan_destroy(vlr_plan* plan);
/* Number of result columns (n_events + 2) and of engine events (1 + 2*n_events). */
int  vlr_plan_n_out(const vlr_plan* plan);
int  vlr_plan_n_samples(const vlr_plan* plan);

/* LDS budget knob: maximum pileup depth per sample the kernel reserves coefficient space for
 * (default 200 = the reference's max_depth default, src/variants/sample.rs:236).  Loci whose kept
 * observations exceed n_samples * depth are reported with VLR_LOCUS_TOO_DEEP.                           */
int  vlr_plan_set_max_depth(vlr_plan* plan, int per_sample_depth);

/* Finer form of the same knob: maximum number of kept observations of ONE locus summed over its samples.  A
 * caller that knows its batch (it decoded the pileups) sets this to the batch maximum: less LDS per
 * workgroup = more resident waves.  vlr_batch_run_host does this automatically.                          */
int  vlr_plan_set_max_obs(vlr_plan* plan, int max_obs_per_locus);

/* The same knob set FROM the batch: `obs_offset_host` are the n_loci * n_samples + 1 pileup offsets of the batch in host memory.
 * Picks the deepest locus — or, where a slightly smaller budget lets sixteen workgroups share a CU (the launcher then runs the
 * kernel build with four waves per SIMD), that budget, provided no locus exceeds it or at most 0.5 % of a batch of 100 000 loci or
 * more do; those take the deep launch (coefficients in the plan's HBM pool) like every locus above the LDS budget.  Returns the
 * budget (>= 1) or an error (< 0).  vlr_batch_run_host applies the same rule per chunk; callers of vlr_batch_run with device
 * batches call this.  (The reference has no such knob: its per-sample depth limit is preprocess's --max-depth, sample.rs:236.) */
int  vlr_plan_fit_max_obs(vlr_plan* plan, const uint32_t* obs_offset_host, int64_t n_loci);

/* Evaluate a batch of loci on the plan's device.  All pointers in `in` / `out` are device pointers.
 * `stream` is a hipStream_t (NULL = default stream); the call is stream-ordered and does not synchronise.
 * Replaces the per-record Caller::call_record (calling.rs:720-842) for n_loci records.                  */
int  vlr_batch_run(vlr_plan* plan, const vlr_batch* in, vlr_results* out, void* stream);

/* Same, but `in` / `out` hold host pointers: stages through device buffers owned by the plan,
 * synchronises before returning.  Still requires the GPU (no CPU fallback).                             */
int  vlr_batch_run_host(vlr_plan* plan, const vlr_batch* in, vlr_results* out);

/* (ABI 4) Device columns in, host results out: `in` holds DEVICE pointers (the batch of a device reader,
 * vlr_obs_table_device_batch), obs_offset_host is the host copy of in->obs_offset (vlr_obs_table_batch), `out` holds HOST
 * pointers.  Only the result buffers are staged; synchronises before returning.                              */
int  vlr_batch_run_device_in(vlr_plan* plan, const vlr_batch* in, const uint32_t* obs_offset_host, vlr_results* out);

/* ------------------------------------------------------------------------------------------------
 * One node, several devices (SURVEY.md 8 e): what the batching shim in Caller::call
 * (/root/reference/src/calling/variants/calling.rs:320-455) binds when the host drives N GPUs from ONE process.
 * vlr_node_create compiles the scenario once per device (devices = NULL, n_devices <= 0: every visible device; the same
 * device may be listed twice — two plans, two streams); vlr_node_batch_run_host cuts the host batch into the contiguous
 * blocks of vlr_node_shard_range (ceil(n / G) loci each, input order — the caller has already collapsed breakend groups
 * to their representatives, calling.rs:569-580, so no group straddles a shard), runs vlr_batch_run_host on one thread and
 * plan per device and returns when every record and AFD list sits at its input position in the caller's arrays.  No
 * collective: inside one process the shards write disjoint ranges of host memory (the multi-process harness,
 * varlociraptor_amd/dist.py, reassembles with one RCCL all-gather instead).  Errors: the first failing shard's code, its
 * message prefixed with the device.                                                                   */
typedef struct vlr_gpu_node vlr_gpu_node;
int  vlr_node_create(const vlr_scenario_desc* desc, int n_devices, const int* devices, vlr_gpu_node** out);
void vlr_node_destroy(vlr_gpu_node* node);
int  vlr_node_n_devices(const vlr_gpu_node* node);
int  vlr_node_device(const vlr_gpu_node* node, int shard);               /* HIP device of shard r */
vlr_plan* vlr_node_plan(vlr_gpu_node* node, int shard);                  /* borrowed: per-device knobs, counters */
int  vlr_node_set_max_depth(vlr_gpu_node* node, int per_sample_depth);   /* vlr_plan_set_max_depth on every plan */
int  vlr_node_set_max_obs(vlr_gpu_node* node, int max_obs_per_locus);
int  vlr_node_shard_range(int64_t n_loci, int n_shards, int shard, int64_t* l0, int64_t* l1);  /* pure arithmetic, no device */
int  vlr_node_batch_run_host(vlr_gpu_node* node, const vlr_batch* in, vlr_results* out);

/* Page-locked host memory for the arrays handed to vlr_batch_run_host (columns in, results out).  The staging
 * copies of vlr_batch_run_host are asynchronous; from pageable memory the runtime bounces them through its own
 * pinned buffer (about 14 GB/s here), from memory obtained here they are direct DMA and overlap the kernel of the
 * previous chunk.  The reference-side binding decodes the observation records (calling.rs:720-760) straight into
 * such arrays.  vlr_host_alloc returns NULL on failure (see vlr_last_error); vlr_host_free(NULL) is a no-op. */
void* vlr_host_alloc(size_t bytes);
void  vlr_host_free(void* p);

/* Size the plan's own device buffers (kernel scratch; with_afd != 0: AFD scratch and log) for batches of up to n_loci loci of
 * at most the configured observation count.  vlr_batch_run grows them on demand with hipMalloc/hipFree, which synchronise the
 * device: after vlr_plan_reserve with the largest batch size, vlr_batch_run only enqueues work on the stream.
 * with_afd: 0, or the afd_capacity of the result buffers that will be passed (1 if unknown: the per-entry key buffer is then grown by
 * the first vlr_batch_run).  The AFD log is budgeted (4 GiB per staging slot, VLR_AFD_LOG_BUDGET_MB): larger batches are walked in
 * sub-ranges of loci that share it.  Both staging slots of vlr_batch_run_host are sized. */
int  vlr_plan_reserve(vlr_plan* plan, int64_t n_loci, int with_afd);

/* Duration in milliseconds of the most recent kernel launch sequence of vlr_batch_run on this plan,
 * measured with HIP events on the launch stream (synchronises on the stop event).  For bench.py.       */
int  vlr_plan_last_kernel_ms(vlr_plan* plan, float* ms);

/* Profiling aid: cumulative {pileup-likelihood evaluations, observation terms} executed by the kernels of this
 * plan since creation (or the last reset); synchronises the device.                                     */
int  vlr_plan_work_counters(vlr_plan* plan, unsigned long long* out2, int reset);

/* ------------------------------------------------------------------------------------------------
 * Read-vs-allele pair HMM: ln P(read window | allele) for a batch of pairs.
 * Replaces, batched, PairHMMRealigner::calculate_prob_allele -> bio PairHMM::prob_related
 * (/root/reference/src/variants/evidence/realignment/mod.rs:519-537) with the emission model of
 * realignment/pairhmm.rs:296-455 (match: 1 - P(miscall), mismatch: P(miscall) * 0.3333, insertion: P(miscall),
 * P(miscall) = 10^(-qual/10)), semiglobal in the allele (free start and end gaps, pairhmm.rs:186-205).
 * The caller (Realigner::allele_support, mod.rs:161-424) keeps the edit-distance pre-filter, builds the allele
 * windows (shrink_to_hit, pairhmm.rs:66-72) and normalises ref against alt.
 *
 * Pair p: allele bases x_bases[x_offset[p] .. x_offset[p+1]) (ASCII, case-insensitive), read window
 * y_bases / y_quals [y_offset[p] .. y_offset[p+1]) (ASCII bases, PHRED qualities without offset), at most 128 read bases
 * (EditDistanceCalculation::max_pattern_len, edit_distance.rs:145-147).  max_edit_dist[p] >= 0 restricts the matrix
 * to cells reachable with at most that many edits (band = hit distance + 4, pairhmm.rs:20); < 0 or a NULL array:
 * full matrix.  gap[] = ln { P(gap in x) = prob_insertion_artifact, P(gap in y) = prob_deletion_artifact,
 * P(extend gap in x), P(extend gap in y) } (pairhmm.rs:119-180; -inf = no extension, the default).
 * ln_prob[p] <- result, -inf for an empty sequence, NaN for a read window above 128 bases.            */
typedef struct vlr_realign_batch_desc {
    int64_t n_pairs;
    const uint32_t* x_offset;       /* [n_pairs + 1] */
    const uint8_t*  x_bases;
    const uint32_t* y_offset;       /* [n_pairs + 1] */
    const uint8_t*  y_bases;
    const uint8_t*  y_quals;
    const int32_t*  max_edit_dist;  /* [n_pairs] or NULL */
    double gap[4];
} vlr_realign_batch_desc;

/* Device pointers, stream-ordered, no synchronisation. */
int vlr_realign_batch(int device, const vlr_realign_batch_desc* pairs, double* ln_prob, void* hip_stream);
/* Host pointers: stages the sequences, runs the kernel, returns when ln_prob is filled. */
int vlr_realign_batch_host(int device, const vlr_realign_batch_desc* pairs, double* ln_prob);

/* The `fast` realignment mode (--pairhmm-mode fast): replaces PathHMMRealigner::calculate_prob_allele
 * (/root/reference/src/variants/evidence/realignment/mod.rs:547-678) — the path probability of an optimal edit-distance alignment
 * (transition terms by the previous operation, emissions as above), best over the hit's alignments.  The reference takes the
 * alignments from bio's Myers traceback, which does not specify the choice among co-optimal alignments; ln_prob[p] here is the
 * best path probability over ALL alignments of minimal semiglobal edit distance (equal when unique, an upper bound otherwise).
 * max_edit_dist is not read.  Same conventions as vlr_realign_batch. */
int vlr_realign_fast_batch(int device, const vlr_realign_batch_desc* pairs, double* ln_prob, void* hip_stream);
int vlr_realign_fast_batch_host(int device, const vlr_realign_batch_desc* pairs, double* ln_prob);

/* The `homopolymer` realignment mode (--pairhmm-mode homopolymer, /root/reference/src/cli.rs:912-947): replaces
 * HomopolyPairHMMRealigner::calculate_prob_allele (/root/reference/src/variants/evidence/realignment/mod.rs:680-730) -> bio
 * HomopolyPairHMM::prob_related — the pair HMM of vlr_realign_batch with per-base hop states for homopolymer run errors (the mode
 * the reference's nanopore / PCR-homopolymer testcases run).  hop[16] = ln of HopParams (realignment/pairhmm.rs:207-295), each
 * group in the order A, C, G, T: prob_seq_homopolymer (start a run error in the read), prob_ref_homopolymer (in the allele),
 * prob_seq_extend_homopolymer, prob_ref_extend_homopolymer; -inf = impossible (the reference's default for all sixteen, which
 * makes the mode equal to vlr_realign_batch).  Same batch layout, band and result conventions as vlr_realign_batch. */
int vlr_realign_homopolymer_batch(int device, const vlr_realign_batch_desc* pairs, const double* hop, double* ln_prob, void* hip_stream);
int vlr_realign_homopolymer_batch_host(int device, const vlr_realign_batch_desc* pairs, const double* hop, double* ln_prob);

/* Edit-distance pre-filter of the same pairs: replaces EditDistanceCalculation::calc_best_hit
 * (/root/reference/src/variants/evidence/realignment/edit_distance.rs:164-260, bio Myers find_all_lazy) as far as
 * Realigner::prob_allele and the band of the pair HMM use it.  dist[p] = smallest semiglobal edit distance of the read
 * window against the allele window (free start and end in the allele; bases compared case-insensitively), end[p] = first
 * allele position (exclusive, 1-based) at which an alignment with that distance ends, n_hits[p] = number of such end
 * positions; dist = -1 for an empty sequence or a read window above 128 bases.  `end` and `n_hits` may be NULL.  y_quals,
 * max_edit_dist and gap of `pairs` are not read.  The band of vlr_realign_batch is max_edit_dist[p] = dist[p] + 4. */
int vlr_edit_distance_batch(int device, const vlr_realign_batch_desc* pairs, int32_t* dist, int32_t* end, int32_t* n_hits, void* hip_stream);
int vlr_edit_distance_batch_host(int device, const vlr_realign_batch_desc* pairs, int32_t* dist, int32_t* end, int32_t* n_hits);

/* ------------------------------------------------------------------------------------------------
 * Bayesian FDR control: the threshold search of `filter-calls control-fdr`
 * (/root/reference/src/filtration/fdr.rs:107-141): sort the posterior (ln) probabilities of the chosen events in
 * descending order, PEP = 1 - p (with smart != 0 the probabilities are first converted to 1 - p: they are PROB_ABSENT
 * [+ PROB_ARTIFACT] sums, fdr.rs:96-114), expected FDR = running mean of the PEPs (bio expected_fdr), threshold = probability of
 * the last entry whose expected FDR is <= alpha and whose PEP differs from its predecessor's.  ln_prob: host array of n
 * values (utils::collect_prob_dist, utils/mod.rs:236-270, any order).  *status: VLR_FDR_EMPTY (no values: the reference's
 * None), VLR_FDR_VALUE (*threshold set), VLR_FDR_LN_ONE (already the best entry exceeds alpha: threshold = ln 1),
 * VLR_FDR_NONE (no admissible entry: None).  The filtering pass (utils/mod.rs:288-374) stays with the caller. */
enum { VLR_FDR_EMPTY = 0, VLR_FDR_VALUE = 1, VLR_FDR_LN_ONE = 2, VLR_FDR_NONE = 3 };
int vlr_fdr_threshold(int device, const double* ln_prob, int64_t n, int smart, double alpha_ln, double* threshold, int* status);

/* The whole of `varlociraptor filter-calls control-fdr` (/root/reference/src/filtration/fdr.rs:36-158, record typing
 * src/utils/collect_variants.rs:44-304, probability sums and the filtering pass src/utils/mod.rs:169-374): read the calls BCF
 * `in_path`, collect the posterior of `events` (names as on the command line: PROB_<UPPER CASE> INFO tags; tags the header does
 * not declare are dropped, none left = error "invalid FDR control events") per typed variant, find the threshold on `device`
 * (vlr_fdr_threshold), and write the kept records, byte for byte as they came, with the input's header to the BCF `out_path`.
 * mode: VLR_FDR_MODE_* (global = 0; LOCAL: threshold ln(1 - alpha); SMART: PROB_ABSENT [+ PROB_ARTIFACT] based, fdr.rs:96-114).
 * vartype: NULL = all variants, else "SNV" "MNV" "INS" "DEL" "BND" "INV" "DUP" "REP"; minlen / maxlen: -1 = open (length range
 * [minlen, maxlen) as Variant::is_type).  Records of one breakend EVENT share one decision.  n_kept / n_total may be NULL. */
enum { VLR_FDR_MODE_LOCAL = 1, VLR_FDR_MODE_SMART = 2, VLR_FDR_MODE_RETAIN_ARTIFACTS = 4 };
int vlr_calls_filter_fdr(const char* in_path, const char* out_path, int n_events, const char* const* events, double alpha, uint32_t mode,
                         const char* vartype, int64_t minlen, int64_t maxlen, int device, int n_threads, int64_t* n_kept, int64_t* n_total);

/* Diagnostics: the DEVICE build of the platform-independent decision arithmetic (include/vlr_detmath.h; which = 0 det_exp,
 * 1 det_log1p_pos, 2 det_log2_ratio(a, b), 3 det_exp2) and of the kernel's mantissa logarithm (4), element-wise on host
 * arrays.  tests/test_gpu_math.py requires 0-3 to be bit-identical to the host build of the same header. */
int vlr_selftest_math(int device, int which, const double* a, const double* b, double* out, int64_t n);
/* Diagnostics: `reps` launches of a stream of known size in the engine's own access widths — mode 0 reads n f32 with
 * lane-contiguous 4-byte loads (4 n bytes per launch), mode 1 writes n f64 with lane-contiguous 8-byte stores (8 n bytes
 * per launch) — so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated for them (tools/traffic_measure.sh). */
int vlr_selftest_stream(int device, int mode, int64_t n, int reps);

/* ---------------------------------------------------------------- native ingest / emission
 * The process boundary of `call variants` without htslib: observation files in (calling.rs:306-339 bcf::Reader,
 * preprocessing/mod.rs:818-919 read_observations, utils/mod.rs:449-474 MiniLogProb), calls file out
 * (calling/variants/mod.rs:178-600 Call::write_final_record).  BGZF blocks are inflated / deflated and records decoded /
 * encoded on n_threads host threads (<= 0: all hardware threads, at most 128). */
typedef struct vlr_obs_table vlr_obs_table;

/* Per-locus site data of a table (pointers owned by the table, valid until vlr_obs_table_free). */
typedef struct {
    int64_t n_loci;
    int32_t n_contigs;
    int32_t _pad;
    const char* const* contig_names;     /* [n_contigs]                                                              */
    const int32_t* contig;               /* [n_loci] index into contig_names                                         */
    const int64_t* pos;                  /* [n_loci] 1-based                                                         */
    const char* strings;                 /* NUL-terminated strings addressed by the offsets below                    */
    const uint64_t* id_offset;           /* [n_loci] record ID ("." if none)                                         */
    const uint64_t* ref_offset;          /* [n_loci] REF allele                                                      */
    const uint64_t* alt_offset;          /* [n_loci] ALT alleles, comma separated                                    */
    /* breakend groups (HaplotypeIdentifier, variants/model/mod.rs:87-133; calling.rs:569-580, 726-741): index of the first
     * locus with the same INFO EVENT / (ID, MATEID) pair — that locus is evaluated, the others copy its result        */
    const int64_t* group_representative; /* [n_loci] (= own index for ungrouped records)                             */
    /* variant-specific priors of the candidate record (calling.rs:470-494), natural log, NaN = absent               */
    const double* heterozygosity_ln;
    const double* somatic_effective_mutation_rate_ln;
    const int32_t* third_allele_evidence; /* [n_obs] output-only feature of the OBS string (mod.rs:283-287), -1 = None */
    const uint8_t* imprecise;            /* [n_loci] INFO IMPRECISE                                                  */
    /* (ABI 3) 64-bit hash of the group's identifier, 0 for ungrouped records: group_representative only looks inside one
     * table; a driver that streams a file in chunks (vlr_obs_reader_next) keeps key -> result of the first record and hands
     * it to the later breakends of the event, as the reference does across the whole file (calling.rs:569-580, 726-741) */
    const uint64_t* group_key;           /* [n_loci]                                                                 */
} vlr_obs_sites;

/* Read one observation file per sample (sample-index order; BCF2 in BGZF blocks, gzip or plain; text VCF accepted) into one
 * table: the SoA columns of vlr_batch in page-locked memory (vlr_host_alloc; plain memory when no device is present), pileup
 * p = locus * n_samples + sample.  locus_flags follow WorkItem.check_* (calling.rs:557-598) under the --omit-* mask.
 * Errors: missing format version 15 (calling.rs:324-339), records that differ between the files (calling.rs:369-390),
 * truncated vectors. */
int  vlr_obs_read(int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_table** out);
void vlr_obs_table_free(vlr_obs_table* table);
int  vlr_obs_table_batch(const vlr_obs_table* table, vlr_batch* out);   /* host pointers into the table, for vlr_batch_run_host */
int  vlr_obs_table_sites(const vlr_obs_table* table, vlr_obs_sites* out);
/* write_observations (preprocessing/mod.rs:921-1038) of sample `sample` of a HOST batch as an observation BCF; sites == NULL:
 * contig "1", positions 1.., alleles synthesised from variant_type / ref_base / alt_base.  third_allele_evidence may be NULL. */
int  vlr_obs_write(const char* path, const vlr_batch* in, int sample, const vlr_obs_sites* sites, const int32_t* third_allele_evidence, int n_threads);
/* The calls file for the loci of `table` from HOST results: path ending in ".bcf" -> BCF2 (BGZF), otherwise text VCF.
 * header_text: "##..." lines and the "#CHROM" line with the sample names; out_names[results->n_out]: the names behind the columns
 * of ln_posterior ("absent", the scenario events, "artifact") — INFO PROB_<NAME>, PHRED, f32, sorted by descending probability. */
int  vlr_calls_write(const char* path, const char* header_text, const vlr_obs_table* table, const vlr_results* results,
                     const char* const* out_names, int n_threads);
/* The same in pieces: a reader that delivers at most max_records records of every sample file per call (bounded memory; the
 * caller may overlap the read of the next chunk with the evaluation and emission of the previous one), and a writer that appends
 * one chunk per call.  vlr_obs_reader_next sets *out = NULL once the files are exhausted; every table is freed by the caller.
 * Breakend groups (vlr_obs_sites.group_representative) are formed within a chunk; group_key identifies an event across chunks. */
typedef struct vlr_obs_reader vlr_obs_reader;
int  vlr_obs_reader_open(int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_reader** out);
int  vlr_obs_reader_next(vlr_obs_reader* reader, int64_t max_records, vlr_obs_table** out);
void vlr_obs_reader_close(vlr_obs_reader* reader);
typedef struct vlr_calls_writer vlr_calls_writer;
int  vlr_calls_writer_open(const char* path, const char* header_text, vlr_calls_writer** out);
int  vlr_calls_writer_append(vlr_calls_writer* writer, const vlr_obs_table* table, const vlr_results* results, const char* const* out_names, int n_threads);
int  vlr_calls_writer_close(vlr_calls_writer* writer);   /* header (if nothing was appended), BGZF end-of-file member, close */
/* Several writers, one file (the shards of a sharded run, vlr_obs_reader_open_device_shard): every shard writes its records as a PART —
 * part 0 with the header, the others without (with_header = 0), none with the end-of-file member (with_eof = 0); BGZF members
 * concatenate, so vlr_calls_concat_parts(path, parts, n) copies the parts in shard order behind each other, appends the end-of-file
 * member and removes the part files.  BCF only. */
int  vlr_calls_writer_set_part(vlr_calls_writer* writer, int with_header, int with_eof);
int  vlr_calls_concat_parts(const char* path, const char* const* parts, int n_parts);
/* Measurement aid: wall seconds of the stages of the last vlr_obs_read ([0] file reads, [1] BGZF inflate, [2] parse + decode — summed
 * over the sample files, which run side by side — [3] all files, [4] merge into the table, [5] strings and groups, [6] total) and of
 * the last vlr_calls_write ([8] record encoding, [9] BGZF deflate + file write, [10] total). */
/* Diagnostics: the calls writer's own "%.<digits>f" (digits 0..3) of one value, for the tests to hold against printf. */
int  vlr_selftest_format_fixed(double v, int digits, char* out, int cap);
/* the device formatter of vlr_results.afd_text on host arrays (tests): n_lists lists of `capacity` slots; text / span as in vlr_results;
 * *n_unformatted = lists left to the arrays */
int  vlr_selftest_afd_text(int device, int64_t n_lists, int capacity, const int32_t* count, const double* vaf, const double* lnprob,
                           uint8_t* text, uint64_t text_capacity, uint32_t* span, uint32_t* n_unformatted);
void vlr_ingest_last_timings(double* out16);
/* The same indices summed over every vlr_obs_reader_next / vlr_calls_writer_append call since the last reset (reset != 0 clears
 * them after reading): what the streaming front door spends per stage over a whole file. */
void vlr_ingest_total_timings(double* out16, int reset);

/* ---- Device front door (ABI 4).  The same streaming reader with the BGZF inflate, the record split and the v15 decode as kernels
 * on `device` (csrc/vlr_inflate.hip, csrc/vlr_decode.hip): the compressed members cross PCIe, the inflated records and the SoA
 * columns are born in device memory.  Replaces the same reference rows as vlr_obs_reader_open (bcf::Reader over the observation
 * files, calling.rs:297-339; read_observations, preprocessing/mod.rs:818-919; MiniLogProb, utils/mod.rs:449-474).  BCF2 in BGZF
 * members only (the files `varlociraptor preprocess` writes); anything else: VLR_ERR_UNSUPPORTED, use vlr_obs_reader_open.
 * The tables of such a reader hold the batch twice: in device memory (vlr_obs_table_device_batch: the pointers vlr_batch_run takes,
 * valid until vlr_obs_table_free) and in page-locked host memory (vlr_obs_table_batch: what the calls writer formats DP / SAOBS /
 * SROBS / OBS from), both complete when vlr_obs_reader_next returns.  The CRC32 of every member is checked against its trailer on
 * the device (ABI 5; htslib does the same per block), like the host reader checks it on the CPU. */
int  vlr_obs_reader_open_device(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads, vlr_obs_reader** out);
/* Sharded device reader: N readers (the ranks of a torchrun job, the devices of a node) each inflate and decode about 1 / N of every
 * file instead of all of it.  The reference reads every record once (calling.rs:306-339, 357-367) and the loci shard across the GPUs
 * (north star): a shard is a contiguous range of records, every reader delivers the records of ITS range in file order.
 *   vlr_obs_reader_open_device_shard   inflates the members of the reader's byte share of every file (plus a lead-in and a tail of
 *                                      1/32 of the share, VLR_INGEST_SHARD_SLACK) and finds the record starts in them;
 *   vlr_obs_reader_shard_counts        out[n_samples][vlr_obs_reader_shard_row_size()]: what the other readers need to know (records that
 *                                      start in the share, where its first record starts and where the one behind its last record does);
 *   the caller gathers the rows of all shards (one small all-gather; plain memory inside one process) in shard order and calls
 *   vlr_obs_reader_shard_assign        on every reader: checks that consecutive shards meet in every file (a wrong guess of a record start
 *                                      shows here: VLR_ERR_INVALID_ARGUMENT, read unsharded), numbers the records, positions the files at
 *                                      the reader's range — the records of the FIRST file's share — and reports it (*first_record,
 *                                      *n_records; the ranges of all shards partition the file in shard order).
 * vlr_obs_reader_next then delivers that range like any reader.  Breakend groups and per-variant prior overrides that reach across
 * a shard boundary are the caller's business (the CLI refuses such files in sharded mode). */
int  vlr_obs_reader_open_device_shard(int device, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads,
                                      int shard, int n_shards, vlr_obs_reader** out);
int  vlr_obs_reader_shard_row_size(void);
int  vlr_obs_reader_shard_counts(const vlr_obs_reader* reader, int64_t* out);
int  vlr_obs_reader_shard_assign(vlr_obs_reader* reader, const int64_t* all_rows, int64_t* first_record, int64_t* n_records);
/* The same for a node driven from one process (vlr_node_*): one sharded reader per device of the node, opened side by side, the
 * counts exchanged in memory; out[vlr_node_n_devices(node)].  Reader r delivers the records of shard r from device vlr_node_device(node, r):
 * its tables go to vlr_batch_run_device_in on vlr_node_plan(node, r), its calls to a part writer (vlr_calls_writer_set_part). */
int  vlr_node_obs_readers_open(vlr_gpu_node* node, int n_samples, const char* const* paths, uint32_t omit_bias_mask, int n_threads,
                               vlr_obs_reader** out);
int  vlr_obs_table_device_batch(const vlr_obs_table* table, vlr_batch* out);  /* VLR_ERR_INVALID_ARGUMENT for a table of a host reader */
/* keep == 0: the tables of this device reader do not bring the observation columns down to the host; instead a kernel (one wave per
 * pileup) derives what the calls writer formats from them — the OBS text itself (distinct observation keys counted, ordered and written
 * as in Call::write_final_record, calling/variants/mod.rs:233-360), the Kass-Raftery letters of SAOBS / SROBS, the prob_mapping runs
 * of DP — and only those summaries cross PCIe.  The observation arrays of vlr_obs_table_batch and
 * vlr_obs_sites.third_allele_evidence are then NOT filled until vlr_obs_table_fetch_columns copies them on demand (the writer does
 * that itself where it has to: pileups of more than 1 024 observations).  Default: keep.
 * vlr_obs_table_summaries: 1 when the calls writer will format this table from device summaries, 0 when from the columns;
 * *n_overflow (may be NULL) = pileups the kernel left to the columns. */
int  vlr_obs_reader_set_host_columns(vlr_obs_reader* reader, int keep);
int  vlr_obs_table_summaries(const vlr_obs_table* table, int64_t* n_overflow);
int  vlr_obs_table_fetch_columns(vlr_obs_table* table);
/* on != 0: vlr_obs_reader_next of this device reader returns while the copy of the observation columns to the host side of the
 * table is still in flight (the device side, what vlr_batch_run_device_in evaluates, is complete).  vlr_calls_writer_append and
 * vlr_obs_table_fetch_columns wait for it; a caller that reads the observation arrays of vlr_obs_table_batch itself calls
 * vlr_obs_table_fetch_columns first.  Default: off. */
int  vlr_obs_reader_set_async_columns(vlr_obs_reader* reader, int on);
/* The inflate stage alone, host buffers in and out (tests, tools): `bgzf` is a sequence of BGZF members (SAM spec 4.1), *out_bytes
 * receives the sum of their ISIZE fields (also when out_capacity is too small: VLR_ERR_INVALID_ARGUMENT then). */
int  vlr_bgzf_inflate(int device, const void* bgzf, int64_t n_bytes, void* out, int64_t out_capacity, int64_t* out_bytes);
/* Measurement aid of the device reader: seconds per stage summed since the last reset — [0] file read + member index, [1] H2D of the
 * compressed bytes, [2] inflate kernel, [3] record split, [4] INFO scan, [5] decode, [6] D2H of columns and cold records, [7] host
 * side (cold records, table), [8] total; [9] inflated bytes, [10] compressed bytes, [11] records, [12] chunks whose record split fell
 * back to the serial walk, [13] seconds of the inflate kernels alone (HIP events on their stream, summed over the files). */
void vlr_ingest_device_timings(double* out16, int reset);
/* Closed device readers leave their device buffers (inflate windows, member lists; hundreds of MB per file) parked for the next reader
 * of the process, bounded by VLR_INGEST_PARK_MB (default 2048).  A long-lived process calls this to hand all of it back to the device. */
void vlr_ingest_device_trim(void);

#ifdef __cplusplus
}
#endif
#endif /* VLR_H */
