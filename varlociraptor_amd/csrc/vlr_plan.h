// vlr_plan.h — device-visible plan layout shared by the host plan compiler (vlr_host.cpp) and the
// gfx950 kernels (vlr_kernels.hip).  Not part of the public ABI (that is include/vlr.h).
#pragma once
#include <stdint.h>

namespace vlr {

constexpr int kMaxSamples = 16;       // == VLR_MAX_SAMPLES: size of the per-sample arrays of the plan layout below
#ifndef VLR_LDS_SAMPLES
#define VLR_LDS_SAMPLES 8
#endif
constexpr int kLdsSamples = VLR_LDS_SAMPLES;  // per-sample arrays of the KERNEL's LDS state: 8 in the standard build (the static LDS of a
                                      // workgroup decides the occupancy of the tumor-normal workloads), 16 in the wide build
                                      // (vlr_kernels_wide.hip) the launcher takes for plans with 9..16 samples
// Limits of the STANDARD build of the kernels, whose static LDS decides the occupancy of the tumor-normal workloads, and of the WIDE
// build (vlr_kernels_wide.hip, vlr_kernels_widedeep.hip) that plans beyond them are routed to: sixteen samples, eight l2fc terms and
// eight nested ranges on a path (the reference has no such limits: grammar/vaftree.rs:168-305 builds arbitrary trees).
constexpr int kMaxLfcStd = 4, kMaxLfcWide = 8;                  // LFC terms on one root->leaf path
constexpr int kMaxRangeDepthStd = 4, kMaxRangeDepthWide = 8;    // nested Range levels on one path
#ifdef VLR_WIDE_BUILD
constexpr int kMaxLfc = kMaxLfcWide;
constexpr int kMaxRangeDepth = kMaxRangeDepthWide;
#else
constexpr int kMaxLfc = kMaxLfcStd;
constexpr int kMaxRangeDepth = kMaxRangeDepthStd;
#endif
constexpr int kMaxFrames = 32;        // explicit recursion stack of the VAF-tree walk (LDS sized by the plan's deepest path, checked at plan creation)
constexpr int kTableCap = 128;        // visited points of one range chain the AFD filter stages in LDS (57 at resolution 0.01)
constexpr int kTableCapMax = 1024;    // upper limit of the per-plan table capacity, which is derived from the finest resolution (2 + 3 rounds + 7:
                                      // resolution 1e-40 needs 971); whether the tables of a plan fit the LDS is checked at plan creation
constexpr int kMaxSet = 1024;         // members of one Set spectrum (LDS: 8 B x samples x the plan's largest set; the reference has no limit)
constexpr int kMaxNamedEvents = 30;   // scenario events (engine universe = 1 + 2*named); 1 + named event groups fit the 31 value bits of the int32 alive masks
constexpr int kMaxNamedEventsWide = 62;  // ... of the wide build, whose masks of event groups are 64 bits (low word | the *_hi word of the plan records);
                                         // the reference has no limit (grammar/mod.rs:129-190)
constexpr int kNHyp = 9;              // 0 = Artifacts::none(), 1..8 single-artifact combinations
constexpr int kXcds = 8;             // XCDs of the MI355X, each with its own L2: consecutive workgroup ids go round them (call kernel: XCD-aware locus mapping)
constexpr int kXcdMapMaxSamples = 2;  // ... for plans of up to this many samples (measured: profiles/r06g.md)
constexpr int kRows = 4;             // concurrent innermost chains per wave of the call kernel: one per 16-lane DPP row
constexpr int kLdsWg16 = 163840 / 16;  // LDS bytes of a workgroup (static + dynamic) up to which SIXTEEN workgroups share a CU (tools/budget_probe.py:
                                       // 10 192 B fit, 10 256 B do not — the host rounds the pileup budget up to a multiple of four observations)
constexpr int kCacheWays = 4;         // per-sample pileup-likelihood cache entries
constexpr int kMaxBatchPoints = 16;   // points evaluated by one eval_pileup call
constexpr int kContainStack = 24;     // explicit stack of the VAFTree::contains walk
constexpr int kNVariantTypes = 5;
constexpr int kMaxDLeaf = 256;        // leaves of one all-discrete root evaluated across the lanes (4 per lane)
constexpr int kMaxDKeys = 64;         // distinct (sample, VAF, contaminant VAF) pileup likelihoods of the discrete roots

// hypothesis slots, in the cartesian order of Artifacts::all_artifact_combinations
// (reference src/variants/model/bias/mod.rs:131-218; alt-locus varies fastest)
enum Hyp { H_NONE = 0, H_ALB = 1, H_HE = 2, H_SCB = 3, H_RPB = 4, H_F1R2 = 5, H_F2R1 = 6, H_SBF = 7, H_SBR = 8 };

struct DevSpectrum {
    int32_t kind;  // 0 set, 1 range
    int32_t set_off, set_len;
    int32_t lex, rex;
    int32_t pad;
    double start, end;
};

struct DevNode {
    int32_t kind, sample, sample_b, cmp;
    double lfc_value;
    DevSpectrum vafs;
    int32_t positive;
    int32_t refbase, altbase;
    int32_t child_off, n_children;
    int32_t alive_mask;  // Sample nodes: event groups (bit 0 = absent, 1 + e = event e) with a spectrum for this sample that overlaps
                         // this node's spectrum at all (closed intervals, 1e-9 slack) — no other group can `contain` an operand set
                         // that takes this sample's VAF from here; lets the walk drop cross-event MAP candidates early
    int32_t alive_mask_hi;  // groups 32..62 (read by the wide build only)
    int32_t pad;
};

// All-discrete roots (every node a Sample node with a Set or single-valued spectrum, e.g. the pedigree scenarios and
// the `absent` chain) are flattened on the host: one record per root-to-leaf path, in walk order.
struct DevDLeaf {
    double vaf[kMaxSamples];
    int32_t prior_idx;          // index into one variant type's prior table
    uint32_t posmask;           // samples whose node on this path holds only VAFs > 0 (dead under clear_ref, generic.rs:270-291)
    uint32_t cmask;             // other event groups whose VAF tree contains these operands (vaftree.rs:42-51)
    uint8_t key[kMaxSamples];   // per sample: index of its (VAF, contaminant VAF) pair in DevPlan::dkey
    uint32_t cmask_hi;          // groups 32..62 (read by the wide build only)
};
struct DevDKey { int32_t sample, pad; double a, b; };

// Roots whose walk is a constant of the plan (DESIGN.md §3 "compiled roots"): a chain of single-valued Sample nodes — one per sample
// but one — ending in a leaf Sample node with a proper Range spectrum, no l2fc / Variant nodes (tumor-normal: somatic_tumor,
// germline_het, germline_hom; single-sample scenarios: every Range event).  What GenericPosterior::density does on the way down
// (modes/generic.rs:247-397: clear-ref shortcuts per node, operands, is_discrete flags) only depends on two per-sample flags of
// the locus, so the plan compiler writes the root out as a record and the kernel's event loop builds the chain task from it
// without walking the tree: no frames, no node loads, no per-node alive / prior-class loops.  The general walk remains for
// everything else and is what the records are checked against (VLR_NO_FAST_ROOTS=1 switches them off).
struct DevFastRoot {
    int32_t kind;                  // 0: general walk, 1: chain record below, 2: general walk, known not to be deferrable in the probe pass
    int32_t n_fixed;               // single-valued Sample nodes above the leaf, in path order
    int32_t inner;                 // sample of the leaf Range node
    int32_t leaf_node;
    int32_t alive;                 // other event groups that can still contain the operands at the leaf (static: walk_root's c.alive)
    int32_t disc;                  // is_discrete mask of the fixed samples
    int32_t pidx;                  // prior-table index of the fixed samples' classes (the integrated sample adds its own)
    int32_t lex, rex;              // the leaf's range
    int32_t alive_hi;              // groups 32..62 of `alive` (read by the wide build only)
    double start, end;
    int32_t fsample[kMaxSamples];
    double fvaf[kMaxSamples];
};

// prior "class" per sample: the reference's Prior::compute
// (src/variants/model/prior.rs:298-438,715-762) depends on a VAF only through equality tests with
// k/ploidy and universe membership, so it is tabulated on the host over per-sample classes.
enum PriorKind { PK_UNIFORM = 0, PK_GERMLINE = 1, PK_SOMATIC = 2 };

// The plan header travels BY VALUE in the kernarg segment and its per-sample arrays are read with scalar loads the compiler likes to
// hoist whole: sized for sixteen samples they cost the standard build 60 more spilled SGPRs and 8 % of its speed.  So the header is a
// template over the array size: the host compiles every plan into DevPlanT<16> and hands the standard and the deep build a
// DevPlanT<8> copy (narrow_plan() below), the wide build the original.
template <int NS>
struct DevPlanT {
    int32_t S, n_named, n_univ, absent_root;
    int32_t n_nodes, max_range_depth, table_size, table_cap;
    double resolution[NS];
    double rho[NS];   // purity (1 - contamination fraction); 1 for uncontaminated samples
    double irho[NS];  // 1 - purity
    int32_t by[NS];   // contaminant sample or -1
    int32_t uni_off[NS + 1];
    int32_t prior_kind[NS];
    int32_t ploidy[NS];
    int32_t n_class[NS];
    int32_t class_stride[NS];
    const DevNode* nodes;
    const int32_t* child_index;
    const double* vafs;
    const int32_t* roots;       // root node ids, absent root NOT included
    const int32_t* root_off;    // [n_named + 1]
    const DevSpectrum* universe;
    const double* prior_table;  // [kNVariantTypes][table_size]
    // event groups (0 = absent, 1 + e = scenario event e; an event and its artifact twin share a VAF tree):
    // union of the spectra of all Sample nodes of sample s in the trees of group g, used as a cheap
    // necessary condition before the full VAFTree::contains walk for cross-event MAP candidates
    const int32_t* grp_spec_off;    // [(n_named + 1) * S + 1]
    const DevSpectrum* grp_spec;
    // flattened all-discrete roots: droot[2 * i], droot[2 * i + 1] = leaf range of root i (0 = absent, 1 + k = roots[k]);
    // a start of -1 sends the root through the general walk
    int32_t n_dkey, n_dleaf;
    int32_t max_frames;         // deepest explicit-stack depth of the walk over this plan's trees
    int32_t max_tab_depth;      // visited-point tables needed by NON-leaf Range frames (leaf chains use the row tables)
    int32_t max_set, pad1;      // members of the largest Set spectrum of a Sample node (>= 1)
    const DevDLeaf* dleaf;
    const DevDKey* dkey;
    const int32_t* droot;
    const DevFastRoot* froot;   // [1 + n_roots] like droot (0 = absent, 1 + k = roots[k])
};
using DevPlan = DevPlanT<kLdsSamples>;
inline DevPlanT<8> narrow_plan(const DevPlanT<16>& w) {   // valid for plans of at most eight samples
    DevPlanT<8> n{};
    n.S = w.S; n.n_named = w.n_named; n.n_univ = w.n_univ; n.absent_root = w.absent_root;
    n.n_nodes = w.n_nodes; n.max_range_depth = w.max_range_depth; n.table_size = w.table_size; n.table_cap = w.table_cap;
    for (int s = 0; s < 8; ++s) {
        n.resolution[s] = w.resolution[s]; n.rho[s] = w.rho[s]; n.irho[s] = w.irho[s]; n.by[s] = w.by[s];
        n.prior_kind[s] = w.prior_kind[s]; n.ploidy[s] = w.ploidy[s]; n.n_class[s] = w.n_class[s]; n.class_stride[s] = w.class_stride[s];
    }
    for (int s = 0; s <= 8; ++s) n.uni_off[s] = w.uni_off[s];
    n.nodes = w.nodes; n.child_index = w.child_index; n.vafs = w.vafs; n.roots = w.roots; n.root_off = w.root_off;
    n.universe = w.universe; n.prior_table = w.prior_table; n.grp_spec_off = w.grp_spec_off; n.grp_spec = w.grp_spec;
    n.n_dkey = w.n_dkey; n.n_dleaf = w.n_dleaf; n.max_frames = w.max_frames; n.max_tab_depth = w.max_tab_depth;
    n.max_set = w.max_set; n.pad1 = w.pad1; n.dleaf = w.dleaf; n.dkey = w.dkey; n.droot = w.droot; n.froot = w.froot;
    return n;
}


// SoA observation columns + per-locus columns (device pointers), mirrors vlr_batch
struct DevBatch {
    int64_t n_loci;
    const uint32_t* obs_offset;
    const float *pm, *pa, *pr, *miss, *psa, *pdo, *phb, *hpa, *hpv;
    const uint32_t* flags;
    const uint8_t *locus_flags, *variant_type, *ref_base, *alt_base;
};

struct DevResults {
    double* ln_posterior;  // [n_loci * (n_named + 2)]
    double* ln_marginal;   // nullable
    double* map_vaf;       // [n_loci * S]
    uint8_t* map_bias;     // [n_loci * 6] nullable
    int32_t* best_event;   // nullable
    uint32_t* status;
    unsigned long long* work;  // nullable: [0] pileup evaluations, [1] observation terms (profiling aid)
    // AFD (optional second "replay" launch): is_discrete flags of the MAP operands (internal, written by the first
    // launch), and the caller's AFD buffers
    uint16_t* map_disc;     // [n_loci]
    int32_t* afd_count;     // [n_loci * S]
    double* afd_vaf;        // [n_loci * S * afd_capacity]
    double* afd_lnprob;     // [n_loci * S * afd_capacity]
    long long* afd_key;     // [n_loci * S * afd_capacity] l2fc-list key of every AFD entry (kernel scratch, plan-owned): part of the
                            // reference's map key, so equal VAFs under different l2fc lists stay two entries
    int32_t afd_capacity;
    int32_t replay;         // 0: call pass, 1: AFD replay pass
    double* escratch;       // [n_loci * (2 S + max_obs)] kernel scratch, plan-owned; per locus: 2 S words = (mantissa, exponent) of
                            // prod_i (w A_i + u_i), a sample's pileup likelihood at alpha = beta = 1 formed without the cancellation of c + q + e
                            // (written per hypothesis for the samples that need it), then the third likelihood coefficient of every kept observation
    // AFD log (plan-owned, only when AFD lists are requested): the call pass appends every evaluated leaf operand set of the
    // clean events — whole visited-point tables of the Range chains, single discrete leaves — to a per-locus region of
    // afd_log_stride 8-byte words; vlr_afd_kernel filters it once the MAP is known.  Word 0 of a region = words used, or
    // -1 when the region overflowed (that locus falls back to the replay launch).
    double* afd_log;
    long long afd_log_stride;
    // deep launch (vlr_kernels_deep.hip): pool of coefficient triples for the loci the LDS-resident kernel flagged
    // VLR_LOCUS_TOO_DEEP; deep_used is a bump counter (doubles) reset before every deep launch
    double* deep_pool;
    unsigned long long* deep_used;
    long long deep_capacity;
};

}  // namespace vlr
