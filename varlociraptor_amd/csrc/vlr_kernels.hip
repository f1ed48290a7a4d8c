// vlr_kernels.hip — gfx950 (MI355X, CDNA4) kernel of the per-locus Bayesian likelihood engine.
//
// One wavefront (64 lanes, one workgroup) evaluates one candidate locus end to end:
//   phase A  pileup statistics for bias gating (wave ballots / reductions over the SoA columns)
//   phase B  for every surviving bias hypothesis: per-observation affine coefficients -> LDS, then the
//            VAF-tree walk of every event with the reference's adaptive integration; pileup
//            likelihoods are evaluated as products prod_i (c_i + q_i*alpha + e_i*beta) with the lanes
//            split into (point, observation-slice) groups
//   phase C  posterior normalisation, artifact aggregation, MAP selection
//
// What is computed follows the reference (varlociraptor v8.9.3) function by function; citations are
// file:line under /root/reference/src.  HOW it is computed is not a translation: the reference
// evaluates ~8 log-space transcendentals per (observation, VAF point) through LRU caches; here the
// observation likelihood is affine in the VAF, so coefficients are built once per
// (locus, hypothesis) and a VAF point costs 2 FMAs + a mantissa/exponent product per observation.
//
// No MFMA: this is f64 VALU + LDS work (north star); the HBM traffic is one pass over the columns.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vlr.h"
#include "../../include/vlr_detmath.h"
#include "vlr_plan.h"

#pragma clang fp contract(off)

// Deep build (vlr_kernels_deep.hip includes this file with VLR_DEEP_BUILD and the namespace renamed): the same call kernel with the
// coefficient triples of a locus in a plan-owned HBM pool instead of LDS, launched behind the normal kernel for the loci it
// flagged VLR_LOCUS_TOO_DEEP (pileups above the LDS budget; the reference has no depth limit, sample.rs:236 is only a default).
#ifdef VLR_DEEP_BUILD
#define VLR_DEEP 1
#else
#define VLR_DEEP 0
#endif

namespace vlr {

// optional per-phase cycle accounting (build with -DVLR_PROFILE): wall cycles of the wave spent per phase
#if defined(VLR_PROFILE) && defined(VLR_PROFILE_VALU)
// Per-region VALU INSTRUCTION counts instead of cycles: tools/valu_profile.sh rewrites the kernel's assembly so that every basic
// block adds its number of VALU instructions to s100 (a register the compiler leaves alone: it stops at s99); the regions read it
// where the cycle profile reads the clock.
#define PROF_DECL unsigned long long prof[40]; unsigned long long prof_t;
#define PROF_START(c) do { unsigned t_; asm volatile("s_mov_b32 s100, 0\n\ts_mov_b32 %0, 0" : "=s"(t_)); (c).prof_t = t_; } while (0)
#define PROF_ADD(c, i) do { unsigned t_; asm volatile("s_mov_b32 %0, s100" : "=s"(t_)); (c).prof[i] += (unsigned)(t_ - (unsigned)(c).prof_t); (c).prof_t = t_; } while (0)
#elif defined(VLR_PROFILE)
#define PROF_DECL unsigned long long prof[40]; unsigned long long prof_t;
#define PROF_START(c) (c).prof_t = __builtin_amdgcn_s_memtime()
#define PROF_ADD(c, i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); (c).prof[i] += t_ - (c).prof_t; (c).prof_t = t_; } while (0)
#else
#define PROF_DECL
#define PROF_START(c)
#define PROF_ADD(c, i)
#endif
// Issue priority of a wave (s_setprio): experiment of round 5 — the term products of a pass carry three independent chains per lane and
// tolerate waiting for an issue slot, the control of the integrator around them is one dependent chain.  -DVLR_PRIO=1: products low, the
// rest high; 2: the other way round; unset: no priority instructions.
#if defined(VLR_PRIO) && VLR_PRIO == 1
#define VLR_PRIO_PRODUCTS() __builtin_amdgcn_s_setprio(0)
#define VLR_PRIO_CHAIN() __builtin_amdgcn_s_setprio(3)
#elif defined(VLR_PRIO) && VLR_PRIO == 2
#define VLR_PRIO_PRODUCTS() __builtin_amdgcn_s_setprio(3)
#define VLR_PRIO_CHAIN() __builtin_amdgcn_s_setprio(0)
#else
#define VLR_PRIO_PRODUCTS()
#define VLR_PRIO_CHAIN()
#endif

#define VLR_NEG_INF (-__builtin_huge_val())
__device__ constexpr double kLn05 = -0.6931471805599453;    // ln 0.5   (utils/mod.rs:45 PROB_05)
__device__ constexpr double kLn095 = -0.05129329438755058;  // ln 0.95  (utils/mod.rs:48 PROB_095)
__device__ constexpr double kLn2 = 0.6931471805599453;
__device__ constexpr double kLn20 = 2.995732273553991;   // ln 20: KassRaftery::Strong as a log Bayes factor
__device__ constexpr double kLn3 = 1.0986122886681098;   // ln 3: KassRaftery::Positive
__device__ constexpr double kEps = 2.220446049250313e-16;

// Masks of event groups (bit 0 = `absent`, bit 1 + e = named event e): one int in the standard build (kMaxNamedEvents = 30), 64 bits in
// the wide build (kMaxNamedEventsWide = 62; the plan records carry the high word in their *_hi fields, vlr_plan.h).
#ifdef VLR_WIDE_BUILD
typedef long long alive_t;
#define UNI_A(x) UNI64(x)
#define ALIVE_BIT(g) ((long long)(1ull << (g)))
#define ALIVE_FULL(n) ((long long)((1ull << (n)) - 1ull))
#define ALIVE_OF(lo, hi) ((alive_t)(unsigned long long)(unsigned)(lo) | (alive_t)((unsigned long long)(unsigned)(hi) << 32))
__device__ __forceinline__ int alive_ctz(alive_t m) { return __builtin_ctzll((unsigned long long)m); }
#define NODE_ALIVE(nd) ALIVE_OF((nd).alive_mask, (nd).alive_mask_hi)
#else
typedef int alive_t;
#define UNI_A(x) UNI(x)
#define ALIVE_BIT(g) ((int)(1u << (g)))
#define ALIVE_FULL(n) ((int)((1u << (n)) - 1u))
#define ALIVE_OF(lo, hi) (lo)
__device__ __forceinline__ int alive_ctz(alive_t m) { return __builtin_ctz(m); }
#define NODE_ALIVE(nd) ((nd).alive_mask)
#endif

enum FrameKind { FK_BRANCH = 0, FK_SET = 1, FK_RANGE = 2 };
enum RangePhase { RP_INIT = 0, RP_ROUND = 1, RP_TAIL = 2, RP_SIMPSON = 3 };

struct Frame {
    int kind, node, iter, n;
    double accM, accS;  // streaming ln-sum-exp
    int sv_present, sv_disc, sv_nlfc, sv_contained;
    // slot: RANGE: index into rs[] / the visited-point tables; sv_alive: saved cross-event candidate mask
#ifdef VLR_WIDE_BUILD
    int slot, sv_mute;
    alive_t sv_alive;
#else
    int slot;
    alive_t sv_alive;
    int sv_mute, pad2;
#endif
};

struct RangeSt {
    double lo, hi, res, L, R, vL, vR, mid, first_mid;
    double ostart, oend;  // node spectrum, for the `contains` check of MAP candidates
    double pend[11];
    int have_first, phase, npend, simpson_n, tn, sample, olex, orex, leaf, have_mid;
};

// (kRows = 4 concurrent chains per wave, one per 16-lane DPP row: vlr_plan.h)
constexpr int kRowPts = 11;     // pending points of one chain round (Simpson fall-back: 11)
constexpr int kPass = 3;        // points one lane carries through a pass over its observation slice

struct ChainTask {              // one innermost Range chain (see run_chain_batch)
    double lo, hi, res, ostart, oend, fixed, result, bestJ, bestX;
#ifdef VLR_WIDE_BUILD
    int olex, orex, simpson_n, pidx, contained, haveBest, n;
    alive_t alive;
#else
    int olex, orex, simpson_n, pidx, contained;
    alive_t alive;
    int haveBest, n;
#endif
    int u, group, disc, inner;  // deferred event-level chains: destination slot, event group, is_discrete mask, sample
};

struct BatchOuter {              // outer Range frame whose pending points are evaluated as row-parallel inner chains
    double lo, hi, res, fixed_const;
    int simpson, dead, vary, np, c0, nt, chn, s_in, s_out;
    alive_t alive0;  // alive0: other groups that can contain ANY point of the outer range
};
struct WalkSave {                // walk_root state while the kernel's event loop runs a chain batch on its behalf
    double rv;
    int sp, node, nrange, skip_record;
};

struct WaveSt {
    double ops_vaf[kLdsSamples];
    double lfc_val[kMaxLfc];
    int lfc_a[kMaxLfc], lfc_b[kMaxLfc], lfc_cmp[kMaxLfc];
    int nkeep[kLdsSamples], soff[kLdsSamples];
    unsigned char all_posref[kLdsSamples];  // (bytes: the static LDS of the kernel sits 32 B under an occupancy step for 4 x 60x pileups)
    union {
        struct {  // phase A statistics: dead once the hypotheses are gated (before phase B starts)
            double pos_all[kLdsSamples], pos_major[kLdsSamples], pos_rate[kLdsSamples];
            int all_ref[kLdsSamples], strong_all[kLdsSamples];
            int strong_bias[kLdsSamples][kNHyp];
            int any_strong_alt[kLdsSamples], has_ins[kLdsSamples], has_del[kLdsSamples];
        };
        struct {  // phase B: pending points / values of the row-parallel chains
            double bpend[kRows][kRowPts], bvals[kRows][kRowPts];
        };
    };
    unsigned char cacheN[kLdsSamples];
    double curMapVaf[kLdsSamples];
    int cs_node[kContainStack], cs_mask[kContainStack];
    ChainTask task[kRows];
    ChainTask stash;                  // one held event-level chain that found no free row yet (see the event loop: "held chains")
    double stash_vaf[kLdsSamples];
    BatchOuter bo;
    WalkSave wk;
    unsigned long long work[2];  // [0] pileup evaluations, [1] observation terms (lane 0 adds; profiling aid)
    int fastok;  // bit s: all terms of sample s stay >= 2^-200 under the current hypothesis (4-term renormalisation is safe)
    int vfast;   // bit s: all terms of sample s stay >= 2^-70: 13 of them multiply without any renormalisation (register-resident runner)
    int ehas;    // bit s: some kept observation of sample s has prob_sample_alt != 0, i.e. a non-zero third coefficient e
                 // (phase A); otherwise e == 0 exactly for the whole pileup and the HBM scratch row is neither written nor read
};

// ------------------------------------------------------------------------------------------------
// Diagnosis builds of round 6 (VERDICT r05 "next" #1; tools/exec_assert.sh, tools/trace_diff.py).  Neither is part of a shipped library.
//  -DVLR_DBG_EXEC_ASSERT: every cross-lane operation of the call kernel checks the EXEC mask it runs under against what the site
//     assumes, and every UNI() / uni_d() / ldc() that its operand really is the same on all active lanes; what is seen goes to
//     g_xsite[source line] = {site was executed, executions under partial EXEC, executions that break the site's rule, OR of the
//     disabled lanes}.  Rules: DPP row operations and ds_swizzle inside a row need every 16-lane row all on or all off (a lane
//     whose source is disabled reads 0 under bound_ctrl); whole-wave shuffles and row-leader reads need full EXEC; a lane read needs
//     that lane on; a "uniform" value must be uniform.
//  -DVLR_DBG_TRACE: the wave of ONE locus (vlr_debug_trace_arm) appends (id, source line, EXEC, 64 lane values) records to a device
//     buffer at the TRC() points below; two builds of the same source are compared record by record on the host.
#if (defined(VLR_DBG_EXEC_ASSERT) || defined(VLR_DBG_TRACE)) && !VLR_DEEP && !defined(VLR_WIDE_BUILD)
#define VLR_DBG_OWNER 1   // this translation unit defines the device symbols and the host accessors
#endif
#ifdef VLR_DBG_EXEC_ASSERT
constexpr int kXSites = 8192;
#ifdef VLR_DBG_OWNER   // (the deep and wide builds are translation units of their own: their checks compile to nothing)
__device__ unsigned long long g_xsite[kXSites * 4];
#endif
enum { XK_ROW = 0, XK_WAVE = 1, XK_ANY = 2 };
__device__ __forceinline__ bool xrows_partial(unsigned long long e) {
    bool bad = false;
    for (int r = 0; r < 4; ++r) { const unsigned b = (unsigned)(e >> (16 * r)) & 0xffffu; bad = bad || (b != 0u && b != 0xffffu); }
    return bad;
}
__device__ __forceinline__ void xnote(int line, unsigned long long e, bool viol) {
#if VLR_DEEP || defined(VLR_WIDE_BUILD)
    (void)line; (void)e; (void)viol;
#else
    if (__builtin_amdgcn_mbcnt_hi((unsigned)(e >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)e, 0u)) != 0u) return;  // first active lane
    unsigned long long* g = g_xsite + 4 * (line & (kXSites - 1));
    g[0] = 1ull;
    if (e != ~0ull) { atomicAdd(g + 1, 1ull); atomicOr(g + 3, ~e); }
    if (viol) atomicAdd(g + 2, 1ull);
#endif
}
__device__ __forceinline__ void xchk(int kind, int line) {
    const unsigned long long e = __builtin_amdgcn_read_exec();
    xnote(line, e, kind == XK_ROW ? xrows_partial(e) : kind == XK_WAVE ? e != ~0ull : false);
}
__device__ __forceinline__ void xchk_lane(int l, int line) {
    const unsigned long long e = __builtin_amdgcn_read_exec();
    xnote(line, e, ((e >> (l & 63)) & 1ull) == 0ull);
}
__device__ __forceinline__ void xchk_uni(bool differs, int line) {
    const unsigned long long e = __builtin_amdgcn_read_exec();
    const bool viol = __builtin_amdgcn_ballot_w64(differs) != 0ull;
    xnote(line, e, viol);
}
#define VLR_XCHK(kind, line) xchk(kind, line)
#define VLR_XCHK_LANE(l, line) xchk_lane(l, line)
#define VLR_XCHK_UNI(differs, line) xchk_uni(differs, line)
#else
#define VLR_XCHK(kind, line) ((void)0)
#define VLR_XCHK_LANE(l, line) ((void)0)
#define VLR_XCHK_UNI(differs, line) ((void)0)
#endif
#ifdef VLR_DBG_TRACE
constexpr int kTraceCap = 1 << 16;
#ifdef VLR_DBG_OWNER
__device__ double g_trace_val[(size_t)kTraceCap * 64];
__device__ unsigned long long g_trace_hdr[(size_t)kTraceCap * 2];  // (id << 32) | source line, EXEC
__device__ int g_trace_n;
__device__ long long g_trace_locus = -1;
#endif
__device__ __forceinline__ void trc_rec(long long locus, int id, double v, int line) {
#if !VLR_DEEP && !defined(VLR_WIDE_BUILD)
    if (locus != __hip_atomic_load(&g_trace_locus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned long long e = __builtin_amdgcn_read_exec();
    int pos = 0;
    const bool first = __builtin_amdgcn_mbcnt_hi((unsigned)(e >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)e, 0u)) == 0u;
    if (first) pos = atomicAdd(&g_trace_n, 1);
    pos = __builtin_amdgcn_readfirstlane(pos);
    if (pos >= kTraceCap) return;
    g_trace_val[(size_t)pos * 64 + (threadIdx.x & 63)] = v;
    if (first) { g_trace_hdr[2 * (size_t)pos] = ((unsigned long long)(unsigned)id << 32) | (unsigned)line; g_trace_hdr[2 * (size_t)pos + 1] = e; }
#endif
}
#define TRC(c, id, v) trc_rec((c).locus, (id), (double)(v), __LINE__)
#define TRCB(c, id, v) trc_rec((c).locus, (id), __longlong_as_double((long long)(v)), __LINE__)   // raw 64-bit pattern
#else
#define TRC(c, id, v) ((void)0)
#define TRCB(c, id, v) ((void)0)
#endif
#ifdef VLR_DBG_EXEC_ASSERT
#define VLR_RDLANE(v, l) rdlane_i((v), (l), __LINE__)
#define VLR_SHFL(v, l) shfl_chk((v), (l), __LINE__)
#define VLR_SHFL_XOR(v, m) shfl_xor_chk((v), (m), __LINE__)
__device__ __forceinline__ int rdlane_i(int v, int l, int line) { VLR_XCHK_LANE(l, line); return __builtin_amdgcn_readlane(v, l); }
template <class T> __device__ __forceinline__ T shfl_chk(T v, int l, int line) { VLR_XCHK(XK_WAVE, line); return __shfl(v, l, 64); }
template <class T> __device__ __forceinline__ T shfl_xor_chk(T v, int m, int line) { VLR_XCHK(XK_WAVE, line); return __shfl_xor(v, m); }
#else  // (the plain builtins: the shipped code is the same instruction for instruction with and without these names)
#define VLR_RDLANE(v, l) __builtin_amdgcn_readlane((v), (l))
#define VLR_SHFL(v, l) __shfl((v), (l), 64)
#define VLR_SHFL_XOR(v, m) __shfl_xor((v), (m))
#endif

// one observation row (all ten columns) of lane `i`; rows beyond `end` read as an empty observation
struct ObsRow { uint32_t f; float pm, pa, pr, miss, psa, pdo, phb, hpa, hpv; };
__device__ __forceinline__ ObsRow load_obs_row(const DevBatch& batch, uint32_t i, uint32_t end) {
    ObsRow r{0u, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, __builtin_nanf(""), __builtin_nanf("")};
    if (i < end) {
        r.f = batch.flags[i];
        r.pm = batch.pm[i]; r.pa = batch.pa[i]; r.pr = batch.pr[i]; r.miss = batch.miss[i];
        r.psa = batch.psa[i]; r.pdo = batch.pdo[i]; r.phb = batch.phb[i];
        if (batch.hpa) r.hpa = batch.hpa[i];
        if (batch.hpv) r.hpv = batch.hpv[i];
    }
    return r;
}

// ------------------------------------------------------------------------------------------------
// values that are wave-uniform by construction but live in VGPRs/LDS: moving them to SGPRs lets the compiler
// use scalar branches and scalar (cached) loads of plan data instead of vector loads
#ifdef VLR_NO_UNI  // diagnosis builds: leave uniformity to the compiler's own analysis
#define UNI(x) (x)
__device__ __forceinline__ double uni_d(double v) { return v; }
#define UNI64(x) (x)
#else
#define UNI64(x) __double_as_longlong(uni_d(__longlong_as_double(x)))
#ifdef VLR_DBG_EXEC_ASSERT
#define UNI(x) uni_i_chk((x), __LINE__)
__device__ __forceinline__ int uni_i_chk(int v, int line) {
    const int r = __builtin_amdgcn_readfirstlane(v);
    VLR_XCHK_UNI(v != r, line);
    return r;
}
#else
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
#endif
__device__ __forceinline__ double uni_d(double v, int site = __builtin_LINE()) {
    const double r = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    VLR_XCHK_UNI(__double_as_longlong(v) != __double_as_longlong(r), site);
    return r;
}
#endif

// Ordering of LDS traffic between the lanes of ONE wave (the workgroup is a single wave64): same-wave LDS operations
// execute in program order, so all that is needed is that the COMPILER keeps the stores of some lanes ahead of the loads
// of others.  Variants for the build matrix (tests/test_gpu_build_matrix.py).
// The AFD buffers of a locus are written and read back by ONE wave: workgroup scope orders them (the CU's vector L1 is shared
// by the workgroup).  __threadfence() is agent scope, which on this multi-XCD part means an L2 write-back per call — it made
// the AFD pass as expensive as the likelihood evaluation itself.
#define VLR_WG_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")

// VLR_SYNC(): the places that were written as workgroup barriers; the workgroup being one wave, they need no more than the
// compiler-level ordering of VLR_WAVE_FENCE() (a hardware barrier also drains every outstanding load of the wave first)
#if defined(VLR_WB_SYNC)
#define VLR_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); } while (0)
#elif defined(VLR_WB_PLAIN)
#define VLR_WAVE_FENCE() __builtin_amdgcn_wave_barrier()
#else
#define VLR_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
#define VLR_SYNC() VLR_WAVE_FENCE()

// The lane id as a value the optimiser cannot hoist: without it every lane-derived constant of the chain runners
// ((double)(lane - 1), row masks, ...) is computed once before the hypothesis loop and then SPILLED across it.
// (VLR_FRESH_MASK: diagnosis builds keep the opaque id only at the call sites whose bit is set — the sites are numbered in source order)
#ifndef VLR_FRESH_MASK
#define VLR_FRESH_MASK 0xffffffffu
#endif
__device__ __forceinline__ int fresh_lane(int lane) {
#ifdef VLR_DBG_NO_FRESH_LANE
    return lane;
#endif
#ifdef VLR_FRESH_BPERMUTE   // the lane's own id through the LDS crossbar: opaque to the optimiser without inline assembly
    return __builtin_amdgcn_ds_bpermute(lane << 2, lane);
#endif
    asm volatile("" : "+v"(lane));
    return lane;
}

// A wave-uniform double parked in SGPRs (park_sd, at its definition) and re-defined in SGPRs at the point of use (fresh_sd):
// instruction selection otherwise copies an SGPR value that feeds a vector select into VGPRs where it is DEFINED (outside the
// loops), and that copy then lives — spilled to scratch — across the kernel.
// Both statements are EMPTY asm blocks since round 6: they only pin values in registers, every instruction around them is the
// compiler's.  Until then they carried instructions, and both were wrong in ways no compiler check can see:
//  * park_sd read the halves with `v_readfirstlane_b32` INSIDE the asm.  gfx950 needs one wait state between a VALU instruction that
//    writes a VGPR and a lane read of that VGPR; the hazard recogniser inserts it for its own lane reads, not for asm text.  Wherever
//    the scheduler put the `v_add_f64` of `1 - forward_rate` directly in front of the asm, the LOW word came back stale (the high
//    word, read one instruction later, was right): the `-O1` / four-waves deviation of rounds 4-5 — every likelihood of a
//    reverse-strand read off by ~2^-24.
//  * fresh_sd moved the pair with two `s_mov_b32` and plain "=s" outputs.  The first move writes %0 before the second reads its
//    input: without early-clobber outputs the compiler may give %0 the register of that input — which it did as soon as the parked
//    pair had been spilled to a VGPR lane and was reloaded into temporaries in front of the statement — and the double came back
//    as (lo, lo): a reverse rate of exactly 0 for 1 - 0.5.  That was the "max-ILP scheduler with fresh_lane" deviation of round 5.
// Found with the per-wave trace of tools/exec_trace_run.py (first differing value: coefficient q of a reverse-strand read, in both
// configurations); tests/test_build_hygiene.py now refuses lane reads and multi-instruction templates without early-clobber
// outputs in this file's asm statements, and both configurations are back in the build matrix.
struct SgprD { int lo, hi; };
__device__ __forceinline__ SgprD park_sd(double v) {
    // (the empty statement hides from the optimiser that the halves are wave-uniform: it folds a lane read of a value it knows to be
    //  uniform, and the VGPR that is left cannot feed fresh_sd's "+s" operands.  The lane reads themselves are the compiler's: it puts
    //  the `s_nop` between them and the VALU instruction that produced v.)
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm volatile("" : "+v"(lo), "+v"(hi));
    SgprD r;
    r.lo = __builtin_amdgcn_readfirstlane(lo);
    r.hi = __builtin_amdgcn_readfirstlane(hi);
    return r;
}
__device__ __forceinline__ double fresh_sd(const SgprD& v) {
    int lo = v.lo, hi = v.hi;
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
}

// x / 3.0, correctly rounded, in three instructions (Markstein: with r = RN(1/3), q0 = RN(x r), the exact residual x - 3 q0 and
// one more FMA give the IEEE quotient; checked against x / 3.0 on 4e8 random doubles).  The compiler's general division is
// eleven instructions with its scaling steps, and the tail of every chain needs two.
__device__ __forceinline__ double div3(double x) {
    const double r = 0.33333333333333331483;  // RN(1/3)
    const double q0 = x * r;
    return __builtin_fma(__builtin_fma(-3.0, q0, x), r, q0);
}

// wave helpers (wave64; one wave per workgroup so VLR_SYNC() is a wave-level LDS fence).
// Reductions over the wave: four DPP steps inside each 16-lane row (quad_perm, quad_perm, row_half_mirror, row_mirror: every
// lane of a row ends with the row's total), then the four row totals are read into SGPRs and combined — the result is
// wave-uniform.  (The __shfl_xor butterflies these replace go through ds_bpermute: six lane-address registers that the compiler
// kept alive — and spilled — across the whole kernel, and six LDS round trips per reduction.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_ROW, site);
#ifdef VLR_DBG_DPP_SHFL  // diagnosis builds: the four permutations of the reductions through ds_bpermute instead of DPP moves of the halves
    const int l_ = (int)__lane_id();
    const int src_ = CTRL == 0xB1 ? (l_ ^ 1) : CTRL == 0x4E ? (l_ ^ 2) : CTRL == 0x141 ? ((l_ & ~7) | (7 - (l_ & 7))) : CTRL == 0x140 ? ((l_ & ~15) | (15 - (l_ & 15)))
                   : CTRL == 0x124 ? ((l_ & ~15) | ((l_ + 4) & 15)) : CTRL == 0x128 ? ((l_ & ~15) | ((l_ + 8) & 15)) : l_;
    if (CTRL == 0xB1 || CTRL == 0x4E || CTRL == 0x141 || CTRL == 0x140 || CTRL == 0x124 || CTRL == 0x128) return __shfl(v, src_, 64);
#endif
    int lo = __double2loint(v), hi = __double2hiint(v);
#ifdef VLR_DBG_DPP_NOP  // diagnosis builds: wait states between whatever wrote the halves and the DPP reads
    asm volatile("s_nop 7" : "+v"(lo), "+v"(hi));
#endif
    // every lane has a valid source under these permutations: bound_ctrl with an undefined `old` lets the compiler write the
    // destination directly instead of copying the source first (three instructions per f64 permute otherwise)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_ROW, site);
#ifdef VLR_DBG_DPP_NOP
    asm volatile("s_nop 7" : "+v"(v));
#endif
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ double lane_d(double v, int l, int site = __builtin_LINE()) {
    VLR_XCHK_LANE(l, site);
#ifdef VLR_DBG_LANE_SHFL
    return __shfl(v, l, 64);
#endif
#ifdef VLR_DBG_DPP_NOP
    { int lo_ = __double2loint(v), hi_ = __double2hiint(v); asm volatile("s_nop 7" : "+v"(lo_), "+v"(hi_)); v = __hiloint2double(hi_, lo_); }
#endif
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_WAVE, site);
    v += dpp_f64<0xB1>(v, site); v += dpp_f64<0x4E>(v, site); v += dpp_f64<0x141>(v, site); v += dpp_f64<0x140>(v, site);
    return (lane_d(v, 0, site) + lane_d(v, 16, site)) + (lane_d(v, 32, site) + lane_d(v, 48, site));
}
__device__ __forceinline__ double wave_max(double v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_WAVE, site);
    v = fmax(v, dpp_f64<0xB1>(v, site)); v = fmax(v, dpp_f64<0x4E>(v, site)); v = fmax(v, dpp_f64<0x141>(v, site)); v = fmax(v, dpp_f64<0x140>(v, site));
    return fmax(fmax(lane_d(v, 0, site), lane_d(v, 16, site)), fmax(lane_d(v, 32, site), lane_d(v, 48, site)));
}
__device__ __forceinline__ int wave_or(int v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_WAVE, site);
    v |= dpp_i32<0xB1>(v, site); v |= dpp_i32<0x4E>(v, site); v |= dpp_i32<0x141>(v, site); v |= dpp_i32<0x140>(v, site);
    return (__builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16)) | (__builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48));
}
__device__ inline int popc64(unsigned long long m) { return __popcll(m); }

// start + step * i of itertools_num::linspace with two roundings (never contracted to an fma: the grid points must be
// bit-identical to the CPU's, a tail point may sit within one ulp of an l2fc boundary)
__device__ __forceinline__ double lin_pt(double start, double step, double i) { return __dadd_rn(start, __dmul_rn(step, i)); }

// approx::relative_eq!(a, b) defaults (epsilon = max_relative = f64::EPSILON)
__device__ inline bool relative_eq(double a, double b) {
    if (a == b) return true;
    if (isinf(a) || isinf(b)) return false;
    double d = fabs(a - b);
    if (d <= kEps) return true;
    return d <= fmax(fabs(a), fabs(b)) * kEps;
}
// bio LogProb::ln_add_exp
__device__ inline double ln_add_exp(double a, double b) {
    double p0 = fmax(a, b), p1 = fmin(a, b);
    if (p0 == VLR_NEG_INF) return VLR_NEG_INF;
    return p0 + log1p(exp(p1 - p0));
}
// streaming ln-sum-exp accumulator: value = M + ln(S), S counts the max term as 1
__device__ inline void lse_add(double& M, double& S, double v) {
    if (v != v) { M = v; S = 1.0; return; }  // NaN poisons
    if (v == VLR_NEG_INF) return;
    if (M != M) return;
    if (M == VLR_NEG_INF) { M = v; S = 1.0; return; }
    // one exponential for both orders: exp(M - v) for v > M and exp(v - M) otherwise are exp(-|v - M|) (the negation is exact)
    const bool up = v > M;
    const double e = exp(up ? M - v : v - M);
    if (up) { S = S * e + 1.0; M = v; }
    else S += e;
}
__device__ inline double lse_value(double M, double S) {
    if (M == VLR_NEG_INF) return VLR_NEG_INF;
    return M + log(S);
}

// ------------------------------------------------------------------------------------------------
// Plan arrays (nodes, spectra, VAF pool, roots) are read-only device memory addressed with wave-uniform indices.
// Reading them through the constant address space makes the loads scalar (s_load into SGPRs, scalar data cache)
// instead of per-lane vector loads that every lane has to wait for.
#define VLR_K4 __attribute__((address_space(4)))
#ifdef VLR_NO_K4
template <class T>
__device__ __forceinline__ T ldc(const T* ptr, int site = 0) { return *ptr; }
#else
template <class T>
__device__ __forceinline__ T ldc(const T* ptr, int site = __builtin_LINE()) {
    // the address is wave-uniform by construction; say so even where the compiler's divergence analysis cannot see it
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    VLR_XCHK_UNI(a != (((unsigned long long)hi << 32) | lo), site);
    return *(const VLR_K4 T*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
#endif
__device__ __forceinline__ DevSpectrum ld_spec(const DevSpectrum* g) {
    DevSpectrum s;
    s.kind = ldc(&g->kind); s.set_off = ldc(&g->set_off); s.set_len = ldc(&g->set_len);
    s.lex = ldc(&g->lex); s.rex = ldc(&g->rex); s.pad = 0;
    s.start = ldc(&g->start); s.end = ldc(&g->end);
    return s;
}
__device__ __forceinline__ DevNode ld_node(const DevNode* g) {
    DevNode n;
    n.kind = ldc(&g->kind); n.sample = ldc(&g->sample); n.sample_b = ldc(&g->sample_b); n.cmp = ldc(&g->cmp);
    n.lfc_value = ldc(&g->lfc_value);
    n.vafs = ld_spec(&g->vafs);
    n.positive = ldc(&g->positive); n.refbase = ldc(&g->refbase); n.altbase = ldc(&g->altbase);
    n.child_off = ldc(&g->child_off); n.n_children = ldc(&g->n_children); n.alive_mask = ldc(&g->alive_mask);
#ifdef VLR_WIDE_BUILD
    n.alive_mask_hi = ldc(&g->alive_mask_hi);
#else
    n.alive_mask_hi = 0;
#endif
    n.pad = 0;
    return n;
}

// ------------------------------------------------------------------------------------------------
// spectra (grammar/formula.rs:1057-1262)
struct RangeV { double start, end; int lex, rex; };
__device__ inline bool range_is_empty(const RangeV& r) { return r.start == r.end && (r.lex || r.rex); }       // 1078-1080
__device__ inline bool range_is_singleton(const RangeV& r) { return r.start == r.end && !(r.lex || r.rex); }  // 1086-1088
__device__ inline bool range_contains(const RangeV& r, double v) {                                            // 1090-1097
    const bool lo = (r.start < v) | ((r.lex == 0) & (r.start == v));
    const bool hi = (r.end > v) | ((r.rex == 0) & (r.end == v));
    return lo & hi;
}
__device__ inline RangeV range_empty() { return RangeV{0.0, 0.0, 1, 1}; }
__device__ inline RangeV range_intersect(const RangeV& a, const RangeV& o) {  // 1131-1168, 1226-1254
    bool eq = a.start == o.start && a.end == o.end && a.lex == o.lex && a.rex == o.rex;
    bool none = !eq && ((a.end < o.start || a.start > o.end) || (a.end <= o.start && (a.rex || o.lex)) ||
                        (a.start >= o.end && (a.lex || o.rex)));
    if (none) return range_empty();
    RangeV r;
    r.start = fmax(a.start, o.start);
    r.end = fmin(a.end, o.end);
    r.lex = (a.start > o.start) ? a.lex : (a.start < o.start) ? o.lex : (a.lex || o.lex);
    r.rex = (a.end < o.end) ? a.rex : (a.end > o.end) ? o.rex : (a.rex || o.rex);
    return r;
}
__device__ inline double observable_max(const RangeV& r, int n) {  // 1198-1216
    double dn = (double)n;
    if (n < 10 || !(dn * (r.end - r.start) > 1.0)) return r.end;
    double c = dn * r.end;
    if (r.rex && c == floor(c)) c -= 1.0;  // (c % 1.0 == 0.0 of the reference: c is an integer; fmod is a loop on this target)
    c = floor(c);
    if (c == 0.0) return r.end;
    return floor(c) / dn;
}
__device__ inline double observable_min(const RangeV& r, int n) {  // 1170-1196
    double dn = (double)n;
    double min_vaf;
    if (n < 10 || !(dn * (r.end - r.start) > 1.0)) {
        min_vaf = r.start;
    } else {
        double c = dn * r.start;
        if (r.lex && c == floor(c)) {
            double adjusted_end = observable_max(r, n);
            double s1 = ceil(c + 1.0) / dn;
            if (s1 <= 1.0 && s1 <= adjusted_end) return s1;
            double s0 = ceil(c) / dn;
            if (s0 <= 1.0 && s0 <= adjusted_end) return s0;
        }
        min_vaf = ceil(c) / dn;
    }
    if (min_vaf >= observable_max(r, n)) return r.start;
    return min_vaf;
}
__device__ inline bool spectrum_contains(const DevSpectrum& sp, const double* pool, double v) {  // 1035-1040
    if (sp.kind == 0) {
        for (int i = 0; i < sp.set_len; ++i)
            if (ldc(pool + sp.set_off + i) == v) return true;
        return false;
    }
    RangeV r{sp.start, sp.end, sp.lex, sp.rex};
    return range_contains(r, v);
}

// LFC predicates (utils/log2_fold_change.rs)
__device__ inline bool lfc_is_true(int cmp, double value, double a, double b) {  // 17-26, 41-52
    double lfc = vlr_det::det_log2_ratio(a, b);  // exact on power-of-two ratios (include/vlr_detmath.h)
    switch (cmp) {
        case VLR_CMP_EQUAL: return relative_eq(lfc, value);
        case VLR_CMP_GREATER: return lfc > value;
        case VLR_CMP_GREATER_EQUAL: return lfc >= value;
        case VLR_CMP_LESS: return lfc < value;
        case VLR_CMP_LESS_EQUAL: return lfc <= value;
        default: return !relative_eq(lfc, value);
    }
}
__device__ inline RangeV lfc_bounds_of(int cmp, double value, double vaf) {  // 56-93
    double proj = vaf / vlr_det::det_exp2(value);
    if (proj < 0.0 || proj > 1.0) return range_empty();
    switch (cmp) {
        case VLR_CMP_EQUAL: return RangeV{proj, proj, 0, 0};
        case VLR_CMP_GREATER: return RangeV{0.0, proj, 0, 1};
        case VLR_CMP_GREATER_EQUAL: return RangeV{0.0, proj, 0, 0};
        case VLR_CMP_LESS: return RangeV{proj, 1.0, 1, 0};
        case VLR_CMP_LESS_EQUAL: return RangeV{proj, 1.0, 0, 0};
        default: return RangeV{0.0, 1.0, 0, 0};
    }
}
__device__ inline void lfc_invert(int& cmp, double& value) {  // 95-122
    switch (cmp) {
        case VLR_CMP_GREATER: cmp = VLR_CMP_LESS_EQUAL; value = -value; break;
        case VLR_CMP_GREATER_EQUAL: cmp = VLR_CMP_LESS; value = -value; break;
        case VLR_CMP_LESS: cmp = VLR_CMP_GREATER_EQUAL; value = -value; break;
        case VLR_CMP_LESS_EQUAL: cmp = VLR_CMP_GREATER; value = -value; break;
        default: break;
    }
}
__device__ inline bool iupac_contains(int code, int base) {  // grammar/formula.rs:23-43
    if (base == code) return true;
    switch (code) {
        case 'R': return base == 'A' || base == 'G';
        case 'Y': return base == 'C' || base == 'T';
        case 'S': return base == 'G' || base == 'C';
        case 'W': return base == 'A' || base == 'T';
        case 'K': return base == 'G' || base == 'T';
        case 'M': return base == 'A' || base == 'C';
        case 'B': return base == 'C' || base == 'G' || base == 'T';
        case 'D': return base == 'A' || base == 'G' || base == 'T';
        case 'H': return base == 'A' || base == 'C' || base == 'T';
        case 'V': return base == 'A' || base == 'C' || base == 'G';
        case 'N': return true;
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------------
// observation features
struct ObsF {
    double pm, pa, pr;
    uint32_t f;
    bool valid;
};
__device__ inline int f_strand(uint32_t f) { return (f >> VLR_F_STRAND_SHIFT) & 3; }
__device__ inline int f_orient(uint32_t f) { return (f >> VLR_F_ORIENT_SHIFT) & 3; }
__device__ inline int f_altlocus(uint32_t f) { return (f >> VLR_F_ALTLOCUS_SHIFT) & 3; }
__device__ inline int f_hplen(uint32_t f) { return (f & VLR_F_HP_LEN_VALID) ? (int)(int8_t)((f >> VLR_F_HP_LEN_SHIFT) & 0xff) : 0; }

// is_bias_evidence == prob_alt(obs) != ln 0 for the artifact component of hypothesis h
// (bias/mod.rs:52-54; strand_bias.rs:30-54; read_orientation_bias.rs:18-32; read_position_bias.rs:18-25;
//  softclip_bias.rs:15-25; alt_locus_bias.rs:63-84)
__device__ inline bool bias_evidence(int h, uint32_t f, bool has_alt_loci) {
    switch (h) {
        case H_ALB: return has_alt_loci ? (f_altlocus(f) == VLR_ALTLOCUS_MAJOR) : !(f & VLR_F_MAX_MAPQ);
        case H_HE: return f_hplen(f) != 0;  // homopolymer_error.rs:82-84
        case H_SCB: return (f & VLR_F_SOFTCLIPPED) != 0;
        case H_RPB: return (f & VLR_F_READPOS_MAJOR) != 0;
        case H_F1R2: return f_orient(f) != VLR_ORIENT_F2R1;
        case H_F2R1: return f_orient(f) != VLR_ORIENT_F1R2;
        case H_SBF: return f_strand(f) == VLR_STRAND_FORWARD || f_strand(f) == VLR_STRAND_NONE;
        case H_SBR: return f_strand(f) == VLR_STRAND_REVERSE || f_strand(f) == VLR_STRAND_NONE;
        default: return true;
    }
}

// exp(ln_sum_exp(v_i)) over a wave-distributed set, mirroring bio's LogProb::ln_sum_exp formula
// m + ln1p(sum_{i != imax} exp(v_i - m)) (used by strand_bias.rs:80-109, read_position_bias.rs:68-113)
// exp(ln_sum_exp(v_i)) over a wave-distributed set (strand_bias.rs:80-109, read_position_bias.rs:68-113): the maximum is
// found first (pass 1), then every lane accumulates its exp(v - m) terms in double-double and the lane sums are
// combined by a butterfly — the rounded sum does not depend on the order (include/vlr_detmath.h), so the thresholds
// that sit exactly on k/n boundaries (prob_mapping is constant within a pileup) are decided as on the CPU.
struct DdAcc {
    vlr_det::dd s;
};
__device__ __forceinline__ void ddacc_add(DdAcc& a, double v, double m, bool on) {
    if (on && v != VLR_NEG_INF) a.s = vlr_det::dd_add(a.s, vlr_det::det_exp(v - m));
}
__device__ __forceinline__ void ddacc_add_e(DdAcc& a, double e, bool on) {  // e = det_exp(v - m) of a finite v
    if (on) a.s = vlr_det::dd_add(a.s, e);
}
__device__ inline double ddacc_exp(const DdAcc& a, double m) {
    vlr_det::dd t = a.s;
    { vlr_det::dd u{dpp_f64<0xB1>(t.hi), dpp_f64<0xB1>(t.lo)}; t = vlr_det::dd_add_dd(t, u); }
    { vlr_det::dd u{dpp_f64<0x4E>(t.hi), dpp_f64<0x4E>(t.lo)}; t = vlr_det::dd_add_dd(t, u); }
    { vlr_det::dd u{dpp_f64<0x141>(t.hi), dpp_f64<0x141>(t.lo)}; t = vlr_det::dd_add_dd(t, u); }
    { vlr_det::dd u{dpp_f64<0x140>(t.hi), dpp_f64<0x140>(t.lo)}; t = vlr_det::dd_add_dd(t, u); }
    // the four row sums (wave-uniform from here on); a 106-bit sum: the order does not reach the rounded value
    const vlr_det::dd r0{lane_d(t.hi, 0), lane_d(t.lo, 0)}, r1{lane_d(t.hi, 16), lane_d(t.lo, 16)};
    const vlr_det::dd r2{lane_d(t.hi, 32), lane_d(t.lo, 32)}, r3{lane_d(t.hi, 48), lane_d(t.lo, 48)};
    t = vlr_det::dd_add_dd(vlr_det::dd_add_dd(r0, r1), vlr_det::dd_add_dd(r2, r3));
    return vlr_det::exp_lse_from_sum(m, t);
}

// ------------------------------------------------------------------------------------------------
// pileup likelihood: ln prod_i (c_i + q_i*alpha + e_i*beta) at np points, lanes = (point, slice).
// Coefficients are AoS triples {c,q,e} (24 B per observation) so one address serves all three reads;
// consecutive slices read consecutive triples: bank = 6k mod 64, conflict free per 32-lane half.

// mantissa/exponent renormalisation of a POSITIVE NORMAL double with integer ops on the high word (the
// fast path guarantees every partial product stays far above 2^-1022)
__device__ __forceinline__ void renorm_pos(double& P, int& E) {
#ifdef VLR_DBG_RENORM_FREXP  // diagnosis builds (tools/o1_variant.sh): the same renormalisation through the frexp instructions
    int e_;
    P = __builtin_frexp(P, &e_);
    E += e_;
    return;
#endif
    const int hi = __double2hiint(P);
    E += (int)(((unsigned)hi >> 20) & 0x7ffu) - 1022;
    P = __hiloint2double((hi & 0x800fffff) | 0x3fe00000, __double2loint(P));
}

// Partial products of NP points over this lane's observation slice: lane k of a W-lane group (W = 16: one DPP
// row = one chain; W = 64: the whole wave) takes terms k, k+W, k+2W, ...  Every coefficient pair {c, q} is loaded from
// LDS once and used for all NP points.  The third coefficient e only matters where beta != 0 (a VAF of exactly 1 in the
// sample or its contaminant): it lives in a per-locus HBM scratch row (`ecoef`, 8 B per observation, read back by the
// same wave) so that the LDS footprint of a workgroup stays at 16 B per observation (occupancy: see tools/lds_sweep.py);
// USE_E is wave-uniform.  `fast`: every term of this pileup is >= 2^-200 at every VAF (checked when the coefficients
// were built), so four terms are multiplied before one renormalisation and no clamping at zero is needed.
// (written and read by the same wave: workgroup-scope coherence through the CU's vector L1 is enough, the barrier after
// the coefficient pass orders the stores before the loads)
__device__ __forceinline__ double ld_e(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <int NP, int W, bool USE_E>
__device__ __forceinline__ void accum_terms_e(const double* __restrict__ coef, const double* __restrict__ ecoef, int D, int k, bool fast,
                                              const double* al, const double* be, double* P, int* E) {
    constexpr int ST = 2 * W;
    const double* a = coef + 2 * k;
    const double* g = ecoef + k;
    int base = 0;
    if (fast) {
        if (NP >= 3) {  // three or four points per lane: groups of two terms keep the register footprint of the hot loop down
            for (; base + 2 * W <= D; base += 2 * W, a += 2 * ST, g += 2 * W) {
                const double c0 = a[0], q0 = a[1];
                const double c1 = a[ST], q1 = a[ST + 1];
                double e0 = 0.0, e1 = 0.0;
                if (USE_E) { e0 = ld_e(g); e1 = ld_e(g + W); }
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    double L0 = __builtin_fma(q0, al[j], c0), L1 = __builtin_fma(q1, al[j], c1);
                    if (USE_E) { L0 = __builtin_fma(e0, be[j], L0); L1 = __builtin_fma(e1, be[j], L1); }
                    P[j] *= L0 * L1;
                    renorm_pos(P[j], E[j]);
                }
            }
        } else
        for (; base + 4 * W <= D; base += 4 * W, a += 4 * ST, g += 4 * W) {
            const double c0 = a[0], q0 = a[1];
            const double c1 = a[ST], q1 = a[ST + 1];
            const double c2 = a[2 * ST], q2 = a[2 * ST + 1];
            const double c3 = a[3 * ST], q3 = a[3 * ST + 1];
            double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
            if (USE_E) { e0 = ld_e(g); e1 = ld_e(g + W); e2 = ld_e(g + 2 * W); e3 = ld_e(g + 3 * W); }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                double L0 = __builtin_fma(q0, al[j], c0), L1 = __builtin_fma(q1, al[j], c1);
                double L2 = __builtin_fma(q2, al[j], c2), L3 = __builtin_fma(q3, al[j], c3);
                if (USE_E) {  // same operation order as e*beta + (q*alpha + c)
                    L0 = __builtin_fma(e0, be[j], L0); L1 = __builtin_fma(e1, be[j], L1);
                    L2 = __builtin_fma(e2, be[j], L2); L3 = __builtin_fma(e3, be[j], L3);
                }
                P[j] *= (L0 * L1) * (L2 * L3);
                renorm_pos(P[j], E[j]);
            }
        }
        if (base < D) {  // <= 3 remaining slots, the last one partially filled
            double acc[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[j] = 1.0;
            for (; base < D; base += W, a += ST, g += W) {
                const bool v = base + k < D;
                const double* aa = v ? a : coef;
                double c0 = aa[0], q0 = aa[1];
                double e0 = 0.0;
                if (USE_E) { e0 = ld_e(v ? g : ecoef); e0 = v ? e0 : 0.0; }
                c0 = v ? c0 : 1.0; q0 = v ? q0 : 0.0;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    double L0 = __builtin_fma(q0, al[j], c0);
                    if (USE_E) L0 = __builtin_fma(e0, be[j], L0);
                    acc[j] *= L0;
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) { P[j] *= acc[j]; renorm_pos(P[j], E[j]); }
        }
    } else {
        for (; base < D; base += W, a += ST, g += W) {  // robust path: per-term mantissa/exponent split (terms may be denormal or zero)
            const bool v = base + k < D;
            const double* aa = v ? a : coef;
            double c0 = aa[0], q0 = aa[1];
            double e0 = 0.0;
            if (USE_E) { e0 = ld_e(v ? g : ecoef); e0 = v ? e0 : 0.0; }
            c0 = v ? c0 : 1.0; q0 = v ? q0 : 0.0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                double L0 = __builtin_fma(q0, al[j], c0);
                if (USE_E) L0 = __builtin_fma(e0, be[j], L0);
                L0 = L0 < 0.0 ? 0.0 : L0;
                int e;
                const double m = __builtin_frexp(L0, &e);
                P[j] *= m;  // mantissas in [0.5,1): no underflow below ~1000 terms per lane
                E[j] += e;
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) { int e; P[j] = __builtin_frexp(P[j], &e); E[j] += e; }
    }
}
template <int NP, int W>
__device__ __forceinline__ void accum_terms(const double* __restrict__ coef, const double* __restrict__ ecoef, int D, int k, bool fast,
                                            const double* al, const double* be, double* P, int* E) {
#ifdef VLR_DBG_ROBUST_TERMS  // diagnosis builds: every product through the per-term frexp path
    fast = false;
#endif
    bool nz = false;
#pragma unroll
    for (int j = 0; j < NP; ++j) nz = nz || (be[j] != 0.0);
    if (ecoef != nullptr && __ballot(nz)) accum_terms_e<NP, W, true>(coef, ecoef, D, k, fast, al, be, P, E);
    else accum_terms_e<NP, W, false>(coef, ecoef, D, k, fast, al, be, P, E);
}
// product over the W lanes of the group; every lane ends with the result (mantissa in [0.5,1) or 0, exponent)
template <int NP, int W>
__device__ __forceinline__ void reduce_terms(double* P, int* E) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        P[j] *= dpp_f64<0xB1>(P[j]); E[j] += dpp_i32<0xB1>(E[j]);      // quad_perm [1,0,3,2]
        P[j] *= dpp_f64<0x4E>(P[j]); E[j] += dpp_i32<0x4E>(E[j]);      // quad_perm [2,3,0,1]
        P[j] *= dpp_f64<0x141>(P[j]); E[j] += dpp_i32<0x141>(E[j]);    // row_half_mirror
        P[j] *= dpp_f64<0x140>(P[j]); E[j] += dpp_i32<0x140>(E[j]);    // row_mirror
        if (W == 64) {  // the four row products, combined from SGPRs (wave-uniform result)
            P[j] = (lane_d(P[j], 0) * lane_d(P[j], 16)) * (lane_d(P[j], 32) * lane_d(P[j], 48));
            E[j] = (VLR_RDLANE(E[j], 0) + VLR_RDLANE(E[j], 16)) + (VLR_RDLANE(E[j], 32) + VLR_RDLANE(E[j], 48));
        }
        int e2;
        P[j] = __builtin_frexp(P[j], &e2);  // product of <= 64 mantissas >= 2^-64: one renormalisation suffices
        E[j] += e2;
    }
}
template <int W>
__device__ __forceinline__ void accum_terms_n(int cnt, const double* __restrict__ coef, const double* __restrict__ ecoef, int D, int k, bool fast,
                                              const double* al, const double* be, double* P, int* E) {
    switch (cnt) {
        case 1: accum_terms<1, W>(coef, ecoef, D, k, fast, al, be, P, E); break;
        case 2: accum_terms<2, W>(coef, ecoef, D, k, fast, al, be, P, E); break;
        case 3: accum_terms<3, W>(coef, ecoef, D, k, fast, al, be, P, E); break;
        default: accum_terms<4, W>(coef, ecoef, D, k, fast, al, be, P, E); break;
    }
}
template <int W>
__device__ __forceinline__ void reduce_terms_n(int cnt, double* P, int* E) {
    switch (cnt) {
        case 1: reduce_terms<1, W>(P, E); break;
        case 2: reduce_terms<2, W>(P, E); break;
        case 3: reduce_terms<3, W>(P, E); break;
        default: reduce_terms<4, W>(P, E); break;
    }
}

// ln pileup likelihood at np <= 4 points (alpha, beta) on all 64 lanes; lane j < np writes res[j]
__device__ __forceinline__ double ln_mantissa(double m);
// ln of what reduce_terms leaves of a pileup product: a mantissa in [1/2, 1), exactly zero (a term that is zero makes the
// likelihood zero) or NaN
__device__ __forceinline__ double ln_product_mantissa(double Pm) { return Pm > 0.0 ? ln_mantissa(Pm) : (Pm == 0.0 ? VLR_NEG_INF : Pm); }
__device__ inline void eval_pileup(const double* __restrict__ coef, const double* __restrict__ ecoef, int D, bool fast, int np, const double* ptA,
                                   const double* ptB, double* res, int lane) {
    double al[4], be[4], P[4];
    int E[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int jj = j < np ? j : np - 1;
        al[j] = ptA[jj]; be[j] = ptB[jj]; P[j] = 1.0; E[j] = 0;
    }
    accum_terms_n<64>(np, coef, ecoef, D, lane, fast, al, be, P, E);
    // reduction over the wave: inside the quads for every point, then lane t of every quad keeps point t and the quads of a row,
    // then the four rows (through the LDS crossbar) are combined for that point alone: lane t < np ends with point t
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < np) {
            P[j] *= dpp_f64<0xB1>(P[j]); E[j] += dpp_i32<0xB1>(E[j]);      // quad_perm [1,0,3,2]
            P[j] *= dpp_f64<0x4E>(P[j]); E[j] += dpp_i32<0x4E>(E[j]);      // quad_perm [2,3,0,1]
        }
    }
    const int tq = lane & 3;
    double Pm = tq == 1 ? P[1] : tq == 2 ? P[2] : tq == 3 ? P[3] : P[0];
    int Em = tq == 1 ? E[1] : tq == 2 ? E[2] : tq == 3 ? E[3] : E[0];
    Pm *= dpp_f64<0x124>(Pm); Em += dpp_i32<0x124>(Em);                    // row_ror:4
    Pm *= dpp_f64<0x128>(Pm); Em += dpp_i32<0x128>(Em);                    // row_ror:8
    Pm *= VLR_SHFL_XOR(Pm, 16); Em += VLR_SHFL_XOR(Em, 16);
    Pm *= VLR_SHFL_XOR(Pm, 32); Em += VLR_SHFL_XOR(Em, 32);
    {
        int e2;
        Pm = __builtin_frexp(Pm, &e2);  // product of <= 64 mantissas >= 2^-64: one renormalisation suffices
        Em += e2;
    }
    // (Pm is a mantissa in [1/2, 1) after the reduction, or exactly zero: a term that is zero makes the likelihood zero)
    if (lane < np) res[lane] = ln_product_mantissa(Pm) + (double)Em * kLn2;
}

// ------------------------------------------------------------------------------------------------
// sort the visited points of one chain by x (rank sort in LDS) and integrate:
// LogProb::ln_trapezoidal_integrate_grid_exp over the sorted grid (utils/adaptive_integration.rs:133-140).
// Duplicate x (HashMap key collisions in the reference) give zero-width segments = ln 0 terms.
__device__ inline double integrate_table(const double* tx, const double* tv, int n, double* sx, double* sv, int lane) {
    for (int i = lane; i < n; i += 64) {
        double x = tx[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            double y = tx[j];
            rank += (y < x) || (y == x && j < i);
        }
        sx[rank] = x;
        sv[rank] = tv[i];
    }
    VLR_SYNC();
    // the trapezoid in the linear domain relative to the largest value, regrouped per grid point as in the row epilogue of
    // run_chain_batch: sum_seg (e_k + e_{k+1}) (x_{k+1} - x_k)/2 = sum_k e_k (x_{k+1} - x_{k-1})/2, one-sided at the ends — one
    // exponential per point and one logarithm instead of ln_add_exp + ln per segment
    // (tables of any length: lane l takes entries l, l + 64, ...; the same additions in the same order as the two-entry version
    //  this replaces for n <= 128)
    bool nan = false;
    double m = VLR_NEG_INF;
    for (int k = lane; k < n; k += 64) {
        const double v = sv[k];
        nan = nan || (v != v);
        m = fmax(m, v == v ? v : VLR_NEG_INF);
    }
    const unsigned long long anynan = __ballot(nan);
    const double M = wave_max(m);
    double r;
    if (anynan) r = __builtin_nan("");
    else if (M == VLR_NEG_INF || n < 2) r = VLR_NEG_INF;
    else {
        double s = 0.0;
        for (int k = lane; k < n; k += 64) {
            const double xl = sx[k > 0 ? k - 1 : 0], xr = sx[k + 1 < n ? k + 1 : k];
            const double v = sv[k];
            const double t = (v == VLR_NEG_INF ? 0.0 : exp(v - M) * ((xr - xl) / 2.0));
            s = (k == lane) ? t : s + t;
        }
        s = wave_sum(s);
        r = M + log(s);
    }
    VLR_SYNC();
    return r;
}

// ------------------------------------------------------------------------------------------------
struct Ctx {
    const DevPlan* __restrict__ plan;
    WaveSt* w;
    double* coef;                    // AoS coefficient triples {c,q,e} (LDS)
    double* setv;                    // [S][kMaxSet] Set candidates per sample (LDS)
    const double* ecoef;               // third coefficient of every kept observation of this locus (HBM scratch row)
    double *cacheA, *cacheB, *cacheV;  // [S][kCacheWays] per-sample likelihood cache (LDS)
    double* dkeyV;                     // [n_dkey] pileup likelihoods of the flattened discrete roots, per hypothesis (LDS)
    // kshift(c)[s]: binary exponent taken out of the coefficients of sample s under the current hypothesis (0 unless the pileup
    // needed the scaled coefficient pass, see "underflow rescue"); lives right behind the RangeSt array: no pointer of its own
    Frame* frames;                     // [nframes] explicit recursion stack of walk_root (LDS, sized by the plan's deepest path)
    RangeSt* rs;                       // [nrs] adaptive-integration state per nested Range level (LDS)
    int nframes, nrs;
    int cap;                         // capacity of one visited-point table
    double *rowX, *rowV;             // [kRows][cap] visited-point tables of the row-parallel innermost chains
    double* afd_seen;                // replay: [S][seen_cap] recorded discrete operands: (VAF, l2fc-list key) pairs
    int seen_cap;                    // = max(16, largest Set spectrum of the plan)
    double* mapv;                    // replay: [S] MAP VAF per sample
    int* afd_cnt;                    // AFD filter kernel: [S] entries written so far, in LDS (one wave owns the locus: no global atomic per
                                     // entry); nullptr in the replay launch, which counts in afd_count itself
    int* afd_nseen;                  // replay: [S] discrete VAFs of sample s already recorded (overlapping roots/branches
                                     // visit the same operands; the reference's joint_probs map keeps one entry)
    double* tvaf;                    // [kRows][S] outer operands of the chain tasks
    double *tabX, *tabV, *sx, *sv;   // visited tables [depth][kTableCap], sort scratch
    int lane;
    int S;
    int vt;                          // variant type of the locus
    int has_snv, refbase, altbase;
    // operands state (modes/generic.rs:116-121), wave-uniform
    int present, disc, nlfc, contained;
    alive_t alive;   // event groups (other than the current one) whose tree may still contain the current operands
    int group;   // group of the event being evaluated (0 absent, 1 + e)
    int n_slots;
    double* mapJ; double* mapVaf; int* mapHyp;  // best MAP candidate per slot (LDS)
    int hyp;
    // AFD replay pass (calling.rs:889-928): MAP operands of the first pass, recorded per matching operand
    int replay, mapGroup, mapDisc;
    int defer_ok, deferred, ndef, defer_slot;  // event-level deferral of simple chains into a row-parallel batch
    int need_batch, bt_nt, bt_inner;           // walk_root asks the event loop to run run_chain_batch (single inline site, few live registers)
    int nhold, nstash, hold_inner;             // held event-level chains: nhold of them parked in the TOP rows (tasks kRows-1, ...), one more in WaveSt::stash
    int afd_mute;  // replay: the current path repeats an outer VAF already visited by its chain (duplicate map key)
    int ehas;      // bit s: sample s has non-zero third coefficients (constant per locus, see WaveSt::ehas);
                   // bit 16 + s: under the current hypothesis sample s takes its likelihood at alpha = beta = 1 from `ones` (ones_risk)
    double* lg;    // AFD log region of this locus (DevResults::afd_log) or nullptr
    int lg_pos, lg_cap;  // next free word / capacity; lg_pos < 0: overflowed
    int lg_nrec;         // records written so far (directory entries)
    double marginal;
    int64_t locus;
    const DevResults* outp;
    // MAP candidate of the current (event, hypothesis class)
    double curJ;
    int curHyp;
    unsigned status;
    PROF_DECL
};

// Prior::compute via the host-built class table (prior.rs:715-762; see vlr_host.cpp build_prior_table)
__device__ inline int prior_class(const DevPlan& p, int s, double v) {
    if (p.prior_kind[s] == PK_UNIFORM) {
        bool in = false;
        for (int u = p.uni_off[s]; u < p.uni_off[s + 1]; ++u) in = in || spectrum_contains(ld_spec(p.universe + u), p.vafs, v);
        return in ? (v == 0.0 ? 0 : 1) : 2;
    }
    int pl = p.ploidy[s];
    double dp = (double)pl;
    double k = rint(dp * v);
    bool match;
    if (p.prior_kind[s] == PK_GERMLINE) match = relative_eq(dp * v, k);  // prior.rs:236-240
    else match = pl > 0 ? relative_eq(v - k / dp, 0.0) : (v == 0.0);    // prior.rs:440-456 on vaf - n/ploidy
    return (match && k >= 0.0 && k <= dp) ? (int)k : pl + 1;
}
__device__ inline double prior_of(const Ctx& c, int inner, double x) {
    const DevPlan& p = *c.plan;
    int idx = 0;
    for (int s = 0; s < c.S; ++s) idx += prior_class(p, s, (s == inner) ? x : c.w->ops_vaf[s]) * p.class_stride[s];
    return p.prior_table[c.vt * p.table_size + idx];
}

// alpha/beta of sample s given its own VAF a and its contaminant's VAF b:
// L_i = c_i + q_i*(rho*a + (1-rho)*b) + e_i*(rho*[a==1] + (1-rho)*[b==1])   (likelihood.rs:43-53,86-115,171-220)
__device__ inline void alpha_beta(const DevPlan& p, int s, double a, double b, double& al, double& be) {
    if (p.by[s] >= 0) {
        al = p.rho[s] * a + p.irho[s] * b;
        be = p.rho[s] * (a == 1.0 ? 1.0 : 0.0) + p.irho[s] * (b == 1.0 ? 1.0 : 0.0);
    } else {
        al = a;
        be = (a == 1.0) ? 1.0 : 0.0;
    }
}

// scratch row of the third coefficients of sample s, or nullptr when they are all zero (WaveSt::ehas)
__device__ __forceinline__ int* kshift(const Ctx& c) { return (int*)(c.rs + c.nrs); }
// ln 2 x (exponents taken out of the coefficients of the samples in `mask`): added to every pileup log-likelihood of those samples
__device__ __forceinline__ double kshift_ln(const Ctx& c, int mask) {
#ifdef VLR_NO_RESCUE
    return 0.0;
#endif
    int k = 0;
    for (int s = 0; s < c.S; ++s)
        if ((mask >> s) & 1) k += kshift(c)[s];
    return (double)UNI(k) * kLn2;
}
__device__ __forceinline__ const double* ecoef_of(const Ctx& c, int s, int off) {
    return ((c.ehas >> s) & 1) ? c.ecoef + off : nullptr;
}
// The all-ones point.  At a VAF of exactly 1 in the sample (and in its contaminant) the likelihood of an observation is w A + u
// (likelihood.rs:43-53 with af == 1); the affine form evaluates it as (w R + u) + w (A - R), which cancels w R.  Harmless while
// u = prob_mismapping * e^missed keeps the term above 2^-24 w R (every pair-HMM record: prob_missed_allele is ln((A + R) / 2));
// where an observation has w R > 2^24 (w A + u) the coefficient pass leaves the product of the direct terms of the sample
// (mantissa, exponent; the 2 S words in front of the locus' row of DevResults::escratch) and every evaluation at that point takes it instead of the affine product.
__device__ __forceinline__ bool ones_any(const Ctx& c) { return ((unsigned)c.ehas >> 16) != 0u; }
__device__ __forceinline__ bool ones_risk(const Ctx& c, int s) { return (((unsigned)c.ehas >> (16 + s)) & 1u) != 0u; }
__device__ __forceinline__ bool is_all_ones(const DevPlan& p, int s, double a, double b) { return a == 1.0 && (p.by[s] < 0 || b == 1.0); }
// (the 2 S words in front of the locus' scratch row: no pointer of their own in the context)
__device__ __forceinline__ double* ones_ptr(const Ctx& c) { return const_cast<double*>(c.ecoef) - 2 * c.S; }
__device__ __forceinline__ void ones_load(const Ctx& c, int s, double& Pm, int& E) {
    const double* o = ones_ptr(c);
    Pm = ld_e(o + 2 * s);
    E = (int)ld_e(o + 2 * s + 1);
}
// the partial products of one point over a W-lane group replaced by the direct product: lane 0 of the group carries it
__device__ __forceinline__ void ones_inject(const Ctx& c, int s, int k, double& P, int& E) {
    double Pm; int Em;
    ones_load(c, s, Pm, Em);
    P *= (k == 0) ? Pm : 1.0;
    E += (k == 0) ? Em : 0;
}

// cached single-point pileup likelihood of sample s (stands in for the per-sample LRU caches of
// modes/generic.rs:38-53: the normal sample's likelihood is reused across all tumor VAFs)
__device__ inline double sample_lik_point(Ctx& c, int s, double a, double b) {  // uncached single point on all 64 lanes
    WaveSt* w = c.w;
    double al, be;
    alpha_beta(*c.plan, s, a, b, al, be);
    const int off = UNI(w->soff[s]), D = UNI(w->nkeep[s]);
    double P1[1] = {1.0};
    int E1[1] = {0};
    if (__builtin_expect(ones_any(c), 0) && ones_risk(c, UNI(s)) && is_all_ones(*c.plan, UNI(s), uni_d(a), uni_d(b))) ones_load(c, s, P1[0], E1[0]);
    else {
    accum_terms<1, 64>(c.coef + 2 * off, ecoef_of(c, s, off), D, c.lane, (w->fastok >> s) & 1, &al, &be, P1, E1);
    reduce_terms<1, 64>(P1, E1);
    }
    TRC(c, 120, P1[0]); TRC(c, 121, E1[0]); TRC(c, 122, al); TRC(c, 123, be); TRC(c, 124, s);
    if (c.lane == 0) { w->work[0] += 1; w->work[1] += (unsigned long long)D; }
#ifdef VLR_NO_RESCUE
    return uni_d(ln_product_mantissa(P1[0]) + (double)E1[0] * kLn2);
#else
    return uni_d(ln_product_mantissa(P1[0]) + (double)(E1[0] + kshift(c)[s]) * kLn2);
#endif
}
__device__ inline double sample_lik(Ctx& c, int s, double a, double b) {
    WaveSt* w = c.w;
    const int n = UNI((int)w->cacheN[s]);
    const int lim = n < kCacheWays ? n : kCacheWays;
    {   // all ways probed at once (lane i looks at way i)
        const int wi = c.lane < kCacheWays ? c.lane : 0;
        const bool hit = (c.lane < lim) & (c.cacheA[s * kCacheWays + wi] == a) & (c.cacheB[s * kCacheWays + wi] == b);
        const unsigned long long hm = __ballot(hit);
        if (hm) return uni_d(c.cacheV[s * kCacheWays + __builtin_ctzll(hm)]);
    }
    const double r = sample_lik_point(c, s, a, b);
    int slot = n % kCacheWays;
    VLR_SYNC();
    if (c.lane == 0) {
        c.cacheA[s * kCacheWays + slot] = a;
        c.cacheB[s * kCacheWays + slot] = b;
        c.cacheV[s * kCacheWays + slot] = r;
        w->cacheN[s] = (unsigned char)(n + 1 == 252 ? 248 : n + 1);  // a byte: wraps within the same residue mod kCacheWays
    }
    VLR_SYNC();
    return r;
}

// MAP ordering (oracle map_before): higher joint first; ties: lower hypothesis id, then smaller VAF tuple, then the
// operand set whose first differing is_discrete flag is set
__device__ inline bool disc_before(int nd, int od) {
    const int diff = nd ^ od;
    return diff != 0 && (nd & diff & -diff) != 0;
}
// Two operand sets with the same hypothesis and the same VAF tuple have the same joint probability whatever their is_discrete
// flags (prior and likelihood depend on the VAFs only; a failed l2fc predicate makes it -inf).  The engine evaluates a VAF reached
// as a chain point and as a Set member along different code paths whose results may differ in the last bits, the reference (and
// the oracle) take both from one cache: such a pair is an exact tie there, broken by the flags.  Joints within 1e-9 relative are
// candidates for that test; for distinct VAF tuples the plain comparison stands.
__device__ __forceinline__ bool same_joint(double a, double b) {
    return a == b || (fabs(a - b) <= 1e-9 * fabs(b) && fabs(b) < 1.7e308);
}
__device__ inline void map_consider(Ctx& c, double joint, int inner, double x) {
    joint = uni_d(joint); x = uni_d(x); inner = UNI(inner);
    if (!(joint == joint)) return;
    bool better = joint > c.curJ;
    if (__builtin_expect(c.curHyp >= 0 && same_joint(joint, c.curJ), 0)) {
        if (c.hyp != (c.curHyp & 15)) { if (joint == c.curJ) better = c.hyp < (c.curHyp & 15); }
        else {
            bool same = true;
            for (int s = 0; s < c.S; ++s) {
                double v = (s == inner) ? x : c.w->ops_vaf[s];
                double o = c.w->curMapVaf[s];
                if (v != o) { if (joint == c.curJ) better = v < o; same = false; break; }
            }
            if (same) {  // same operands up to the flags: the flags decide, whatever the last bits of the two evaluations say
                const int nd = (inner >= 0) ? (c.disc & ~(1 << inner)) : c.disc, od = c.curHyp >> 4;
                if (nd != od) {
                    better = disc_before(nd, od);
                    joint = fmax(joint, c.curJ);  // one value in the reference; the slot keeps the larger of the two evaluations
                    c.curJ = joint;
                }
            }
        }
    }
    if (c.curHyp < 0) better = true;
    if (better) {
        c.curJ = joint;
        c.curHyp = c.hyp | (((inner >= 0) ? (c.disc & ~(1 << inner)) : c.disc) << 4);  // hypothesis | is_discrete mask
        VLR_SYNC();
        if (c.lane < c.S) c.w->curMapVaf[c.lane] = (c.lane == inner) ? x : c.w->ops_vaf[c.lane];
        VLR_SYNC();
    }
}

// ---- cross-event MAP candidates -------------------------------------------------------------------
// sample_infos (calling.rs:851-864) takes the MAP from ALL visited operands that the best event's tree
// `contains` (vaftree.rs:42-51) — including operands visited while evaluating a different event (e.g. the
// excluded range start 0.0, visited when n_obs < 10, belongs to `absent`).  `alive` tracks, per pushed VAF,
// which other groups can still contain the operands (necessary condition on the union of their spectra);
// at a leaf the survivors get the full contains walk.
__device__ inline int slot_clean(int g) { return g == 0 ? 0 : 1 + 2 * (g - 1); }
__device__ inline int slot_art(const Ctx& c, int g) { return g == 0 ? c.plan->n_univ : 2 + 2 * (g - 1); }

__device__ inline bool group_may_contain(const DevPlan& p, int g, int s, double v) {
    int o0 = ldc(p.grp_spec_off + g * p.S + s), o1 = ldc(p.grp_spec_off + g * p.S + s + 1);
    for (int i = o0; i < o1; ++i)
        if (spectrum_contains(ld_spec(p.grp_spec + i), p.vafs, v)) return true;
    return false;
}
// `alive` and s are wave-uniform (scalar loop over groups, scalar loads of the group spectra); v may differ per lane
__device__ inline alive_t alive_update(const Ctx& c, alive_t alive, int s, double v) {
    alive_t m = UNI_A(alive);
    s = UNI(s);
    alive_t res = m;
    while (m) {
        int g = alive_ctz(m);
        m &= m - 1;
        const bool may = group_may_contain(*c.plan, g, s, v);
        res = may ? res : (res & ~ALIVE_BIT(g));
    }
    return res;
}
// groups among `alive` whose spectra for sample s can contain ANY point of [lo, hi] (closed, slightly widened: a tail
// point of a chain may exceed the bracket by an ulp): a chain whose interval misses all spectra of the other groups
// needs no per-point alive_update
__device__ inline alive_t alive_restrict(const Ctx& c, alive_t alive, int s, double lo, double hi) {
    const DevPlan& p = *c.plan;
    alive_t m = UNI_A(alive);
    s = UNI(s);
    const double a = lo - 1e-9, b = hi + 1e-9;
    alive_t res = 0;
    while (m) {
        const int g = alive_ctz(m);
        m &= m - 1;
        const int o0 = ldc(p.grp_spec_off + g * p.S + s), o1 = ldc(p.grp_spec_off + g * p.S + s + 1);
        bool hit = false;
        for (int i = o0; i < o1 && !hit; ++i) {
            const DevSpectrum sp = ld_spec(p.grp_spec + i);
            if (sp.kind == 0) {
                for (int k = 0; k < sp.set_len; ++k) {
                    const double v = ldc(p.vafs + sp.set_off + k);
                    hit = hit || (v >= a && v <= b);
                }
            } else hit = sp.start <= b && sp.end >= a;
        }
        if (hit) res |= ALIVE_BIT(g);
    }
    return res;
}
// VAFTree::contains (vaftree.rs:42-51,116-164) of group g for the current operands (sample `inner` at x)
__device__ inline bool group_contains(Ctx& c, int g, int inner, double x, int excl = -1) {
    g = UNI(g); inner = UNI(inner); x = uni_d(x);
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    int r0 = (g == 0) ? 0 : ldc(p.root_off + g - 1), r1 = (g == 0) ? 1 : ldc(p.root_off + g);
    int full = (1 << c.nlfc) - 1;
    bool result = false;
    for (int ri = r0; ri < r1 && !result; ++ri) {
        int sp = 0;
        w->cs_node[0] = (g == 0) ? p.absent_root : ldc(p.roots + ri);
        w->cs_mask[0] = full;
        sp = 1;
        while (sp > 0 && !result) {
            sp--;
            int node = UNI(w->cs_node[sp]), mask = UNI(w->cs_mask[sp]);
            const DevNode nd = ld_node(p.nodes + node);
            bool contained;
            if (nd.kind == VLR_NODE_SAMPLE) {
                if (nd.sample == excl) { result = true; continue; }  // vaftree.rs:124-128: excluded sample => true
                double v = (nd.sample == inner) ? x : w->ops_vaf[nd.sample];
                contained = spectrum_contains(nd.vafs, p.vafs, v);
            } else if (nd.kind == VLR_NODE_LFC) {
                bool found = false;
                for (int i = 0; i < c.nlfc; ++i)
                    if ((mask >> i) & 1)
                        if (w->lfc_a[i] == nd.sample && w->lfc_b[i] == nd.sample_b && w->lfc_cmp[i] == nd.cmp && w->lfc_val[i] == nd.lfc_value) {
                            found = true;
                            mask &= ~(1 << i);
                        }
                contained = found;
            } else contained = (nd.kind != VLR_NODE_FALSE);
            if (!contained) continue;
            if (nd.n_children == 0) { if (mask == 0) result = true; continue; }
            for (int ch = nd.n_children - 1; ch >= 0; --ch) {
                if (sp >= kContainStack) { c.status |= VLR_LOCUS_TABLE_FULL; break; }
                w->cs_node[sp] = ldc(p.child_index + nd.child_off + ch);
                w->cs_mask[sp] = mask;
                sp++;
            }
        }
    }
    return result;
}
// candidate for another group's slot; same ordering as map_consider
__device__ inline void cross_consider(Ctx& c, int g, double joint, int inner, double x) {
    joint = uni_d(joint); x = uni_d(x); inner = UNI(inner);
    if (!(joint == joint)) return;
    int slot = (c.hyp == 0) ? slot_clean(g) : slot_art(c, g);
    double curJ = uni_d(c.mapJ[slot]);
    int curHyp = UNI(c.mapHyp[slot]);
    bool better = curHyp < 0 || joint > curJ;
    if (curHyp >= 0 && same_joint(joint, curJ)) {
        if (c.hyp != (curHyp & 15)) { if (joint == curJ) better = c.hyp < (curHyp & 15); }
        else {
            bool same = true;
            for (int s = 0; s < c.S; ++s) {
                double v = (s == inner) ? x : c.w->ops_vaf[s];
                double o = c.mapVaf[slot * c.S + s];
                if (v != o) { if (joint == curJ) better = v < o; same = false; break; }
            }
            if (same) {
                const int nd = (inner >= 0) ? (c.disc & ~(1 << inner)) : c.disc, od = curHyp >> 4;
                if (nd != od) {
                    better = disc_before(nd, od);
                    joint = fmax(joint, curJ);
                    if (!better && joint > curJ) { VLR_SYNC(); if (c.lane == 0) c.mapJ[slot] = joint; VLR_SYNC(); }
                }
            }
        }
    }
    if (better) {
        VLR_SYNC();
        if (c.lane == 0) { c.mapJ[slot] = joint; c.mapHyp[slot] = c.hyp | (((inner >= 0) ? (c.disc & ~(1 << inner)) : c.disc) << 4); }
        if (c.lane < c.S) c.mapVaf[slot * c.S + c.lane] = (c.lane == inner) ? x : c.w->ops_vaf[c.lane];
        VLR_SYNC();
    }
}
// ---- AFD log (DevResults::afd_log): what FORMAT/AFD needs from the call pass (calling.rs:889-928) is the joint probability of
// every visited operand set; which of them end up in the lists is only known once the MAP is.  Re-evaluating the clean events
// in a second launch (the replay below) costs as much as the call itself, so the call pass writes the values it has anyway:
//   record = header word | S outer operands | nlfc x 2 words (l2fc terms on the path) | payload
//   header: n (16 bits) | integrated sample + 1 (5) | is_discrete mask (16) | event group (8) | nlfc (4) | kind (2)
//   kind 1: the visited-point table of a Range chain: n x values, then n joint values;  kind 2: one discrete leaf: its joint
__device__ __forceinline__ bool log_on(const Ctx& c) { return c.lg != nullptr && c.hyp == 0 && !c.replay && c.lg_pos >= 0; }
__device__ __forceinline__ long long log_header(int kind, int n, int s_in, int disc, int group, int nlfc) {
    return (long long)n | ((long long)(s_in + 1) << 16) | ((long long)(disc & 0xffff) << 21) | ((long long)(group & 0xff) << 37) |
           ((long long)(nlfc & 0xf) << 45) | ((long long)kind << 49);
}
// Region layout: word 0 = words used (-1: overflow), word 1 = number of records, words 2 .. 2 + kLogDir = start of every
// record (so that vlr_afd_kernel can look at all headers at once), records from word kLogFirst on.
constexpr int kLogDir = 64;
constexpr int kLogFirst = 2 + kLogDir;
// reserve one record of `words` (uniform); returns its start or -1 after marking the region as overflowed
__device__ __forceinline__ int log_reserve(Ctx& c, int words) {
    if (c.lg_pos + words > c.lg_cap || c.lg_nrec >= kLogDir) { c.lg_pos = -1; return -1; }
    const int at = c.lg_pos;
    if (c.lane == 0) c.lg[2 + c.lg_nrec] = __longlong_as_double((long long)at);
    c.lg_nrec += 1;
    c.lg_pos += words;
    return at;
}
// header + operands + l2fc terms of the CURRENT context (walk state); returns the payload position or -1
__device__ inline int log_begin(Ctx& c, int kind, int n, int s_in, int disc, int payload_words) {
    const int S = c.S, nl = c.nlfc;
    const int at = log_reserve(c, 1 + S + 2 * nl + payload_words);
    if (at < 0) return -1;
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 0) & 1u) ? fresh_lane(c.lane) : (c.lane));  // keeps the lane-derived offsets below out of registers that live across the kernel
    if (lane == 0) c.lg[at] = __longlong_as_double(log_header(kind, n, s_in, disc, c.group, nl));
    if (lane < S) c.lg[at + 1 + lane] = w->ops_vaf[lane];
    if (lane < nl) {
        c.lg[at + 1 + S + 2 * lane] = __longlong_as_double((long long)w->lfc_a[lane] | ((long long)w->lfc_b[lane] << 8) | ((long long)w->lfc_cmp[lane] << 16));
        c.lg[at + 1 + S + 2 * lane + 1] = w->lfc_val[lane];
    }
    return at + 1 + S + 2 * nl;
}
__device__ inline void log_leaf(Ctx& c, double joint) {
    const int at = log_begin(c, 2, 1, -1, c.disc, 1);
    if (at >= 0 && c.lane == 0) c.lg[at] = joint;
}
__device__ inline void log_table(Ctx& c, int s_in, const double* tx, const double* tv, int n) {  // a single chain, all 64 lanes
    const int at = log_begin(c, 1, n, s_in, c.disc & ~(1 << s_in), 2 * n);
    if (at < 0) return;
    for (int i = (((VLR_FRESH_MASK >> 1) & 1u) ? fresh_lane(c.lane) : (c.lane)); i < n; i += 64) { c.lg[at + i] = tx[i]; c.lg[at + n + i] = tv[i]; }
}

// ---- all-discrete roots (DevDLeaf): every leaf of the root on its own lane --------------------------------------
// GenericPosterior::density of a root whose nodes are all Set / single-valued Sample nodes (modes/generic.rs:294-330):
// ln_sum_exp over the leaves of prior + likelihood; a leaf below a node that is dead under clear_ref (270-291) is not
// visited.  The MAP candidates of the root's own slot and of the other groups containing a leaf (static, DevDLeaf::cmask)
// are the arg-best leaves under the map_consider ordering.  Values of the likelihood terms come from c.dkeyV.
__device__ inline bool dleaf_tuple_before(const DevDLeaf* leaves, int la, int lb, int S) {
    for (int s = 0; s < S; ++s) {
        const double va = leaves[la].vaf[s], vb = leaves[lb].vaf[s];
        if (va != vb) return va < vb;
    }
    return false;
}
// wave arg-best of per-lane candidates (J desc, then VAF tuple asc); returns the leaf index (uniform) or -1
__device__ inline int dleaf_wave_best(const DevDLeaf* leaves, double bJ, int bL, int S, double& Jout) {
    const double Jm = wave_max(bL >= 0 ? bJ : VLR_NEG_INF);
    unsigned long long tie = __ballot(bL >= 0 && bJ == Jm);
    int best = -1;
    while (tie) {
        const int ln = __builtin_ctzll(tie);
        tie &= tie - 1;
        const int l = VLR_RDLANE(bL, ln);
        if (best < 0 || dleaf_tuple_before(leaves, l, best, S)) best = l;
    }
    Jout = Jm;
    return best;
}
__device__ inline double eval_discrete_root(Ctx& c, int l0, int l1) {
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 2) & 1u) ? fresh_lane(c.lane) : (c.lane)), S = c.S;
    const DevDLeaf* leaves = p.dleaf;
    unsigned cr = 0;
    for (int s = 0; s < S; ++s)
        if (UNI(w->nkeep[s]) > 10 && UNI((int)w->all_posref[s])) cr |= 1u << s;
    const double* ptab = p.prior_table + (size_t)c.vt * p.table_size;
    constexpr int T = kMaxDLeaf / 64;
    const int TT = (l1 - l0 + 63) >> 6;  // leaves per lane (uniform)
    double jv[T];
    unsigned cm[T];
#ifdef VLR_WIDE_BUILD
    unsigned cmh[T];
    unsigned gmh = 0;
#endif
    bool ok[T];
    bool sawnan = false;
    unsigned gm = 0;
    double bJ = VLR_NEG_INF;
    int bL = -1;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int l = l0 + lane + 64 * t;
        jv[t] = VLR_NEG_INF; cm[t] = 0; ok[t] = false;
#ifdef VLR_WIDE_BUILD
        cmh[t] = 0;
#endif
        if (t < TT && l < l1) {
            const DevDLeaf& L = leaves[l];
            if ((L.posmask & cr) == 0) {
                double lik = 0.0;
                for (int s = 0; s < S; ++s) lik += c.dkeyV[L.key[s]];
                const double joint = ptab[L.prior_idx] + lik;
                if (joint != joint) sawnan = true;
                else {
                    jv[t] = joint; cm[t] = L.cmask; ok[t] = true;
                    gm |= L.cmask;
#ifdef VLR_WIDE_BUILD
                    cmh[t] = L.cmask_hi; gmh |= L.cmask_hi;
#endif
                    if (bL < 0 || joint > bJ || (joint == bJ && dleaf_tuple_before(leaves, l, bL, S))) { bJ = joint; bL = l; }
                }
            }
        }
    }
    if (__ballot(sawnan)) { c.status |= VLR_LOCUS_NAN; return __builtin_nan(""); }
    if (log_on(c)) {  // AFD log: every visited leaf is a complete, all-discrete operand set; one record (kind 3) for the root
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (t < TT) cnt += __popcll(__ballot(ok[t]));
        const int rec = S + 1;
        const int at0 = cnt ? log_reserve(c, 1 + rec * cnt) : -1;
        if (at0 >= 0) {
            if (lane == 0) c.lg[at0] = __longlong_as_double(log_header(3, cnt, -1, (1 << S) - 1, c.group, 0));
            int base = at0 + 1;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t < TT) {
                    const unsigned long long m = __ballot(ok[t]);
                    if (ok[t]) {
                        const int at = base + rec * __popcll(m & ((1ull << lane) - 1ull));
                        const DevDLeaf& L = leaves[l0 + lane + 64 * t];
                        for (int s = 0; s < S; ++s) c.lg[at + s] = L.vaf[s];
                        c.lg[at + S] = jv[t];
                    }
                    base += rec * __popcll(m);
                }
            }
        }
    }
    // density
    double m = VLR_NEG_INF;
#pragma unroll
    for (int t = 0; t < T; ++t) m = fmax(m, jv[t]);
    const double M = wave_max(m);
    double dens = VLR_NEG_INF;
    if (M != VLR_NEG_INF) {
        double ssum = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (t < TT) ssum += ok[t] ? exp(jv[t] - M) : 0.0;
        dens = uni_d(M + log(wave_sum(ssum)));
    }
    // MAP candidates: all operands are discrete
    const int saved_disc = c.disc;
    c.disc = (1 << S) - 1;
    {
        double J;
        const int best = dleaf_wave_best(leaves, bJ, bL, S, J);
        if (best >= 0) {
            VLR_SYNC();
            if (lane < S) w->ops_vaf[lane] = leaves[best].vaf[lane];
            VLR_SYNC();
            map_consider(c, J, -1, 0.0);
        }
    }
    // other groups
    gm = (unsigned)wave_or((int)gm);
#ifdef VLR_WIDE_BUILD
    alive_t gmm = ALIVE_OF(gm, wave_or((int)gmh));
#else
    alive_t gmm = (alive_t)gm;
#endif
    while (gmm) {
        const int g = alive_ctz(gmm);
        gmm &= gmm - 1;
        double gJ = VLR_NEG_INF;
        int gL = -1;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int l = l0 + lane + 64 * t;
#ifdef VLR_WIDE_BUILD
            const bool in_g = g < 32 ? ((cm[t] >> g) & 1u) != 0u : ((cmh[t] >> (g - 32)) & 1u) != 0u;
#else
            const bool in_g = ((cm[t] >> g) & 1u) != 0u;
#endif
            if (t < TT && ok[t] && in_g) {
                if (gL < 0 || jv[t] > gJ || (jv[t] == gJ && dleaf_tuple_before(leaves, l, gL, S))) { gJ = jv[t]; gL = l; }
            }
        }
        double J;
        const int best = dleaf_wave_best(leaves, gJ, gL, S, J);
        if (best >= 0) {
            VLR_SYNC();
            if (lane < S) w->ops_vaf[lane] = leaves[best].vaf[lane];
            VLR_SYNC();
            cross_consider(c, g, J, -1, 0.0);
        }
    }
    c.disc = saved_disc;
    return dens;
}

// AFD replay (calling.rs:889-928): record (VAF of sample s, posterior density) of every clean operand set that
// equals the MAP on all other samples (allele_freq, artifacts, is_discrete) and is contained in the best event
// with sample s excluded.
// The l2fc terms on the path are part of the reference's map key (LikelihoodOperands::lfcs, modes/generic.rs:116-121): two operand
// sets with equal VAFs and flags but different l2fc lists are two entries of joint_probs and appear twice in an AFD list.  A 64-bit
// hash of the ordered list stands for it (0 = no terms).
__device__ inline long long lfc_ctx_key(const Ctx& c) {
    unsigned long long h = 0;
    for (int i = 0; i < c.nlfc; ++i) {
        const unsigned long long t = (unsigned long long)(unsigned)c.w->lfc_a[i] | ((unsigned long long)(unsigned)c.w->lfc_b[i] << 8) |
                                     ((unsigned long long)(unsigned)c.w->lfc_cmp[i] << 16) | ((unsigned long long)(i + 1) << 24);
        h = (h ^ t) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h = (h ^ (unsigned long long)__double_as_longlong(c.w->lfc_val[i])) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
    }
    return c.nlfc ? (long long)(h | 1ull) : 0ll;
}
__device__ inline void afd_consider(Ctx& c, double joint, int inner, double x, int skip_sample = -1) {
    if (c.hyp != 0 || c.afd_mute) return;
    int mism = 0, ms = -1;
    const int disc = (inner >= 0) ? (c.disc & ~(1 << inner)) : c.disc;
    for (int s = 0; s < c.S; ++s) {
        double v = (s == inner) ? x : c.w->ops_vaf[s];
        bool eq = v == c.mapv[s] && (((disc >> s) & 1) == ((c.mapDisc >> s) & 1));
        if (!eq) { mism++; ms = s; }
    }
    if (mism >= 2) return;
    const long long lkey = UNI64(lfc_ctx_key(c));
    for (int s = 0; s < c.S; ++s) {
        if (mism == 1 && s != ms) continue;
        if (s == skip_sample) continue;  // vlr_afd_kernel has written this sample's entries of the record in bulk
        if (!group_contains(c, c.mapGroup, inner, x, s)) continue;
        const double vs = (s == inner) ? x : c.w->ops_vaf[s];
        if ((disc >> s) & 1) {  // discrete operand for s: the whole operand set is a repeat if this (VAF, l2fc list) was recorded before
            int ns = c.afd_nseen[s];
            bool seen = false;
            for (int i = 0; i < ns; ++i)
                seen = seen || (c.afd_seen[2 * (s * c.seen_cap + i)] == vs && __double_as_longlong(c.afd_seen[2 * (s * c.seen_cap + i) + 1]) == lkey);
            if (seen) continue;
            VLR_SYNC();
            if (c.lane == 0 && ns < c.seen_cap) {
                c.afd_seen[2 * (s * c.seen_cap + ns)] = vs; c.afd_seen[2 * (s * c.seen_cap + ns) + 1] = __longlong_as_double(lkey);
                c.afd_nseen[s] = ns + 1;
            }
            VLR_SYNC();
        }
        int idx_l = 0;
        if (c.afd_cnt) {
            idx_l = UNI(c.afd_cnt[s]);
            VLR_WAVE_FENCE();
            if (c.lane == 0) c.afd_cnt[s] = idx_l + 1;
            VLR_WAVE_FENCE();
        }
        if (c.lane == 0) {
            const DevResults& o = *c.outp;
            int64_t slot = c.locus * c.S + s;
            int idx = c.afd_cnt ? idx_l : atomicAdd(&o.afd_count[slot], 1);
            if (idx < o.afd_capacity) {
                double v = (s == inner) ? x : c.w->ops_vaf[s];
                // the sign bit marks a discrete operand until afd_finish() has removed repeated keys
                if ((disc >> s) & 1) v = __hiloint2double(__double2hiint(v) | (int)0x80000000, __double2loint(v));
                o.afd_vaf[slot * o.afd_capacity + idx] = v;
                o.afd_lnprob[slot * o.afd_capacity + idx] = joint - c.marginal;
                o.afd_key[slot * o.afd_capacity + idx] = lkey;
            }
        }
    }
}

// End of the replay pass: the reference's joint_probs is ONE map per record keyed by the operands, so an operand set
// visited from several events / roots (overlapping events, the same Range point reached along two branches) has a
// single AFD entry.  Entries of sample s are keyed by (VAF, is_discrete, l2fc list) — the other samples equal the MAP by
// construction; the first occurrence is kept.
__device__ inline void afd_finish(Ctx& c) {
    const DevResults& o = *c.outp;
    VLR_WG_FENCE();
    for (int s = 0; s < c.S; ++s) {
        const int64_t slot = c.locus * c.S + s;
        int cnt = 0;
        if (c.afd_cnt) cnt = c.afd_cnt[s];
        else if (c.lane == 0) cnt = atomicAdd(&o.afd_count[slot], 0);
        cnt = UNI(cnt);
        if (cnt <= 0) continue;
        if (c.afd_cnt && c.lane == 0) o.afd_count[slot] = cnt;  // (replaced by the de-duplicated count below unless the list was truncated)
        double* vv = o.afd_vaf + slot * o.afd_capacity;
        double* pp = o.afd_lnprob + slot * o.afd_capacity;
        long long* kk = o.afd_key + slot * o.afd_capacity;
        if (cnt <= o.afd_capacity) {
            int kept = 0;
            for (int base = 0; base < cnt; base += 64) {
                const int i = base + c.lane;
                const bool on = i < cnt;
                const double v = on ? vv[i] : 0.0, pr = on ? pp[i] : 0.0;
                const long long key = __double_as_longlong(v), lk = on ? kk[i] : 0ll;
                bool dup = false;
                for (int j = 0; j < kept; ++j) dup = dup | ((__double_as_longlong(vv[j]) == key) & (kk[j] == lk));  // compacted prefix (earlier chunks)
                for (int j = base; j < base + 64 && j < cnt; ++j) {                                  // earlier entries of this chunk
                    const long long kj = VLR_SHFL(key, j - base), lj = VLR_SHFL(lk, j - base);
                    dup = dup | ((j < i) & (kj == key) & (lj == lk));
                }
                const unsigned long long keep = __ballot(on & !dup);
                const int pos = kept + __popcll(keep & ((1ull << c.lane) - 1ull));
                VLR_WG_FENCE();
                if (on & !dup) { vv[pos] = v; pp[pos] = pr; kk[pos] = lk; }
                VLR_WG_FENCE();
                kept += __popcll(keep);
            }
            if (c.lane == 0) o.afd_count[slot] = kept;
            cnt = kept;
        }
        const int lim = cnt < o.afd_capacity ? cnt : o.afd_capacity;
        VLR_WG_FENCE();
        for (int i = c.lane; i < lim; i += 64) vv[i] = fabs(vv[i]);
    }
}

// the reference's joint_probs is a map keyed by the operands: a VAF visited twice by one chain (tail points that
// coincide with end points, the abandoned-arm point) is one entry.  True if x already occurs in tx[0..n).
__device__ inline bool table_has(const double* tx, int n, double x, int lane) {
    bool hit = (lane < n && tx[lane] == x) || (lane + 64 < n && tx[lane + 64] == x);
    return __ballot(hit) != 0ull;
}

// all MAP bookkeeping for one evaluated operand set
__device__ inline void map_all(Ctx& c, double joint, int inner, double x, bool own_path_contained, alive_t alive) {
    alive = UNI_A(alive);
    if (own_path_contained) map_consider(c, joint, inner, x);
    else if (group_contains(c, c.group, inner, x)) map_consider(c, joint, inner, x);  // contained via another path
    while (alive) {
        int g = alive_ctz(alive);
        alive &= alive - 1;
        if (group_contains(c, g, inner, x)) cross_consider(c, g, joint, inner, x);
    }
}

// GenericLikelihood::compute step 1 (modes/generic.rs:503-509) for a leaf operand set
__device__ inline bool lfcs_ok(const Ctx& c, int inner, double x) {
    for (int i = 0; i < c.nlfc; ++i) {
        int sa = c.w->lfc_a[i], sb = c.w->lfc_b[i];
        double va = (sa == inner) ? x : c.w->ops_vaf[sa];
        double vb = (sb == inner) ? x : c.w->ops_vaf[sb];
        if (!lfc_is_true(c.w->lfc_cmp[i], c.w->lfc_val[i], va, vb)) return false;
    }
    return true;
}

// joint probability of the current operands at a leaf (bio Model::compute closure:
// Prior::compute + GenericLikelihood::compute), single point
__device__ inline double leaf_joint(Ctx& c) {
    double joint;
    PROF_ADD(c, 19);  // walk: descent to a discrete leaf
    if (!lfcs_ok(c, -1, 0.0)) {
        joint = VLR_NEG_INF;
    } else {
        double lik = 0.0;
        for (int s = 0; s < c.S; ++s) {
            int by = c.plan->by[s];
            double a = c.w->ops_vaf[s];
            double b = by >= 0 ? c.w->ops_vaf[by] : 0.0;
            lik += sample_lik(c, s, a, b);
        }
        PROF_ADD(c, 20);  // leaf: likelihoods
        joint = prior_of(c, -1, 0.0) + lik;
        PROF_ADD(c, 21);  // leaf: prior
    }
    joint = uni_d(joint);
    TRC(c, 125, joint);
    if (joint != joint) c.status |= VLR_LOCUS_NAN;
    if (log_on(c)) log_leaf(c, joint);
    if (__builtin_expect(c.replay != 0, 0)) afd_consider(c, joint, -1, 0.0);
    else map_all(c, joint, -1, 0.0, c.contained != 0, c.alive);
    PROF_ADD(c, 22);  // leaf: MAP candidates
    return joint;
}

// ---- adaptive integration state machine (utils/adaptive_integration.rs:25-141)
// after the values of r.pend[] are in the table: advance; returns true when the chain is finished
__device__ inline bool range_advance(Ctx& c, RangeSt& r, const double* tx, const double* tv) {
    if (r.phase == RP_SIMPSON) return true;
    if (r.phase == RP_TAIL) return true;
    if (r.phase == RP_INIT) {
        r.L = r.lo;
        r.R = r.hi;
        r.vL = tv[0];
        r.vR = tv[1];
        r.have_mid = 0;
    } else {  // RP_ROUND: argmax over {left, middle1, middle2, right} (61-94); lowest index wins ties.
        // The round's points were appended as (mid, middle1, middle2), so their values are the last two
        // table entries; left/right values are carried along (the reference looks them up in its HashMap).
        const double x0 = uni_d(r.L), x1 = uni_d(r.pend[1]), x2 = uni_d(r.pend[2]), x3 = uni_d(r.R);
        const double v0 = uni_d(r.vL), v1 = uni_d(tv[r.tn - 2]), v2 = uni_d(tv[r.tn - 1]), v3 = uni_d(r.vR);
        int k = 0;
        double vb = v0;
        if (v1 > vb) { k = 1; vb = v1; }
        if (v2 > vb) { k = 2; vb = v2; }
        if (v3 > vb) { k = 3; }
        const double nl = (k <= 1) ? x0 : (k == 2) ? x1 : x2, vl = (k <= 1) ? v0 : (k == 2) ? v1 : v2;
        const double nr = (k == 0) ? x1 : (k == 1) ? x2 : x3, vr = (k == 0) ? v1 : (k == 1) ? v2 : v3;
        r.L = nl; r.vL = vl;
        r.R = nr; r.vR = vr;
    }
    if ((((r.R - r.L) >= r.res) && r.L < r.R) || !r.have_mid) {
        double mid = (r.R + r.L) / 2.0;
        r.mid = mid;
        r.have_mid = 1;
        if (!r.have_first) { r.first_mid = mid; r.have_first = 1; }
        r.pend[0] = mid;
        r.pend[1] = (mid + r.L) / 2.0;
        r.pend[2] = (r.R + mid) / 2.0;
        r.npend = 3;
        r.phase = RP_ROUND;
        return false;
    }
    // tail: small interval around the optimum (107-131)
    // (the reference's "additional grid point in the initially abandoned arm", adaptive_integration.rs:95-106, is (max + first_middle)/2
    // or (first_middle + min)/2: exactly middle2 or middle1 of the FIRST round, a key its HashMap already holds — re-evaluating it
    // changes nothing, so it is not visited again)
    double lo3 = fmax(r.mid - r.res * 3.0, r.lo);
    double hi3 = fmin(r.mid + r.res * 3.0, r.hi);
    double sa = div3(r.mid - lo3), sb = div3(hi3 - r.mid);  // itertools_num::linspace step, n = 4
    r.pend[0] = lin_pt(lo3, sa, 0.0);
    r.pend[1] = lin_pt(lo3, sa, 1.0);
    r.pend[2] = lin_pt(lo3, sa, 2.0);
    r.pend[3] = lin_pt(r.mid, sb, 1.0);
    r.pend[4] = lin_pt(r.mid, sb, 2.0);
    r.pend[5] = lin_pt(r.mid, sb, 3.0);
    r.npend = 6;
    r.phase = RP_TAIL;
    return false;
}

// final value of a finished chain
__device__ inline double range_finish(Ctx& c, RangeSt& r, const double* tx, const double* tv) {
    if (r.phase == RP_SIMPSON) {  // bio LogProb::ln_simpsons_integrate_exp (modes/generic.rs:367-385)
        int n = r.simpson_n;
        double M = VLR_NEG_INF, S = 0.0;
        for (int i = 1; i < n - 1; ++i) lse_add(M, S, tv[i] + log((double)(2 + (i % 2) * 2)));
        lse_add(M, S, tv[0]);
        lse_add(M, S, tv[n - 1]);
        return lse_value(M, S) + log(r.hi - r.lo) - log((double)(n - 1)) - log(3.0);
    }
    return integrate_table(tx, tv, r.tn, c.sx, c.sv, c.lane);
}

// ------------------------------------------------------------------------------------------------
// Innermost Range chain at a leaf node: the hot loop of the whole engine (>95 % of all pileup evaluations).
// Same algorithm as leaf_joint_batch + range_advance, but the chain state lives in registers, the
// likelihood of samples that do not depend on the integrated VAF and the prior index of the outer samples
// are hoisted out of the rounds, each lane selects its point from uniform registers, and results come
// back through lane broadcasts — no barriers, no LDS traffic besides coefficient reads and table appends.
__device__ __forceinline__ double run_leaf_chain(Ctx& c, RangeSt& rl, double* tx, double* tv) {
    PROF_ADD(c, 3);
#ifdef VLR_PROFILE
    c.prof[11] += 1;
#endif
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 3) & 1u) ? fresh_lane(c.lane) : (c.lane));
    const int inner = UNI(rl.sample);
    const double lo = uni_d(rl.lo), hi = uni_d(rl.hi), res = uni_d(rl.res);
    const RangeV orig{uni_d(rl.ostart), uni_d(rl.oend), UNI(rl.olex), UNI(rl.orex)};
    const int simpson_n = UNI(rl.simpson_n);

    double fixed = 0.0;
    int dep = 0, pidx = 0;
    for (int s = 0; s < c.S; ++s) {
        int by = p.by[s];
        if (s == inner || by == inner) dep |= 1 << s;
        else fixed += sample_lik(c, s, w->ops_vaf[s], by >= 0 ? w->ops_vaf[by] : 0.0);
        if (s != inner) pidx += prior_class(p, s, w->ops_vaf[s]) * p.class_stride[s];
    }
    fixed = uni_d(fixed);
    pidx = UNI(pidx);
    const double dep_shift = kshift_ln(c, dep);
    const double* ptab = p.prior_table + c.vt * p.table_size;
    const int istride = p.class_stride[inner];
    const int ncls = p.n_class[inner];
    const double pr0 = uni_d(ptab[pidx]), pr1 = ncls > 1 ? uni_d(ptab[pidx + istride]) : VLR_NEG_INF, pr2 = ncls > 2 ? uni_d(ptab[pidx + 2 * istride]) : VLR_NEG_INF;

    const alive_t alive_c = c.alive ? alive_restrict(c, c.alive, inner, lo, hi) : 0;  // other groups that can contain a point of this chain
    // prior class of the integrated sample: if one Range spectrum of a uniform-prior universe covers [lo, hi], every
    // point of the chain is inside the universe (class 1, or 0 at exactly 0) — no per-point spectrum walk
    bool cls_fast = false;
    if (p.prior_kind[inner] == PK_UNIFORM)
        for (int u = p.uni_off[inner]; u < p.uni_off[inner + 1]; ++u) {
            const DevSpectrum sp = ld_spec(p.universe + u);
            if (sp.kind == 1) {
                RangeV ur{sp.start, sp.end, sp.lex, sp.rex};
                cls_fast = cls_fast || (range_contains(ur, lo) && range_contains(ur, hi));
            }
        }
    // pending points and their joint values live in two tiny LDS arrays (same-wave LDS ops execute in order;
    // wave_barrier() only stops the compiler from reordering them)
    double* pend = w->bpend[0];  // the row buffers are idle while a single chain runs
    double* vals = w->bvals[0];
    int np, phase, tn = 0;
    VLR_WAVE_FENCE();
    if (simpson_n) {
        double step = (hi - lo) / (double)(simpson_n - 1);
        if (lane < simpson_n) pend[lane] = (lane == 0) ? lo : (lane == simpson_n - 1) ? hi : lin_pt(lo, step, (double)lane);
        np = simpson_n;
        phase = RP_SIMPSON;
    } else {
        if (lane < 2) pend[lane] = lane ? hi : lo;
        np = 2;
        phase = RP_INIT;
    }
    VLR_WAVE_FENCE();
    double L = lo, R = hi, vL = VLR_NEG_INF, vR = VLR_NEG_INF, mid = lo, first_mid = lo;
    bool have_first = false, have_mid = false, failed = false;
    double bestJ = VLR_NEG_INF, bestX = 0.0;
    bool haveBest = false;
    int ndep = 0;
    unsigned dep_terms = 0;

    for (;;) {
        if (tn + np > c.cap) { c.status |= VLR_LOCUS_TABLE_FULL; failed = true; break; }
        // one point per 16-lane DPP row (four points per pass); the row's lanes split the observations and the
        // row leader stores the joint value.  (Cheaper than a 64-lane product for a single chain: the reduction
        // stays inside a row.)
        const int row = lane >> 4, rlane = lane & 15;
        bool nan_seen = false;
        for (int p0 = 0; p0 < np; p0 += 4) {
            const int jp = (p0 + row) < np ? (p0 + row) : np - 1;
            const double xr = pend[jp];
            double P1[1] = {1.0};
            int E1[1] = {0};
            int dm = dep;
            while (dm) {
                int s = __builtin_ctz(dm);
                dm &= dm - 1;
                int by = p.by[s];
                double a = (s == inner) ? xr : w->ops_vaf[s];
                double b = by >= 0 ? ((by == inner) ? xr : w->ops_vaf[by]) : 0.0;
                double al, be;
                alpha_beta(p, s, a, b, al, be);
                if (__builtin_expect(ones_any(c), 0) && ones_risk(c, s) && __ballot(is_all_ones(p, s, a, b))) {
                    // (rows are at different points: the rows at the all-ones point take the direct product, the others their own)
                    double Pt[1] = {1.0};
                    int Et[1] = {0};
                    accum_terms<1, 16>(c.coef + 2 * UNI(w->soff[s]), ecoef_of(c, s, UNI(w->soff[s])), UNI(w->nkeep[s]), rlane, (w->fastok >> s) & 1, &al, &be, Pt, Et);
                    if (is_all_ones(p, s, a, b)) { Pt[0] = 1.0; Et[0] = 0; ones_inject(c, s, rlane, Pt[0], Et[0]); }
                    P1[0] *= Pt[0]; E1[0] += Et[0];
                    { int e2; P1[0] = __builtin_frexp(P1[0], &e2); E1[0] += e2; }
                } else
                accum_terms<1, 16>(c.coef + 2 * UNI(w->soff[s]), ecoef_of(c, s, UNI(w->soff[s])), UNI(w->nkeep[s]), rlane, (w->fastok >> s) & 1, &al, &be, P1, E1);
            }
            reduce_terms<1, 16>(P1, E1);
            const double lik = fixed + (log(P1[0]) + (double)E1[0] * kLn2) + dep_shift;
            double jv;
            if (c.nlfc > 0 && !lfcs_ok(c, inner, xr)) jv = VLR_NEG_INF;
            else {
                int cls = cls_fast ? (xr == 0.0 ? 0 : 1) : prior_class(p, inner, xr);
                jv = (cls == 0 ? pr0 : cls == 1 ? pr1 : cls == 2 ? pr2 : ptab[pidx + cls * istride]) + lik;
            }
            const bool lead = rlane == 0 && (p0 + row) < np;
            nan_seen = nan_seen | (lead && jv != jv);
            if (lead) { tx[tn + p0 + row] = xr; tv[tn + p0 + row] = jv; vals[p0 + row] = jv; }
        }
        if (__ballot(nan_seen)) c.status |= VLR_LOCUS_NAN;
        VLR_WAVE_FENCE();
        // lane j < np looks at point j for the MAP bookkeeping
        const bool owner = lane < np;
        const double x = pend[owner ? lane : np - 1];
        const double joint = vals[owner ? lane : np - 1];

        // MAP candidates (calling.rs:851-864)
        const bool own_in = c.contained && range_contains(orig, x);
        const alive_t al2 = alive_c ? alive_update(c, alive_c, inner, x) : 0;
        const bool slow = __ballot(owner && (!own_in || al2 != 0)) != 0ull;
        (void)joint;
        if (c.replay) {
            for (int i = 0; i < np; ++i) {
                double xi = uni_d(pend[i]);
                if (!table_has(tx, tn + i, xi, lane)) afd_consider(c, uni_d(vals[i]), inner, xi);
            }
        } else if (!slow) {
            for (int i = 0; i < np; ++i) {
                double v = uni_d(vals[i]), xi = uni_d(pend[i]);
                if (v == v && (!haveBest || v > bestJ || (v == bestJ && xi < bestX))) { bestJ = v; bestX = xi; haveBest = true; }
            }
        } else {
            for (int i = 0; i < np; ++i) {
                double xi = pend[i];
                map_all(c, vals[i], inner, xi, c.contained && range_contains(orig, xi), alive_c ? alive_update(c, alive_c, inner, xi) : 0);
            }
        }
        tn += np;

        // ---- advance the chain (utils/adaptive_integration.rs:54-131)
        if (phase == RP_SIMPSON || phase == RP_TAIL) break;
        if (phase == RP_INIT) {
            L = lo; R = hi; vL = uni_d(vals[0]); vR = uni_d(vals[1]);
        } else {  // argmax over {left, middle1, middle2, right}; lowest index wins ties
            double xs0 = L, xs1 = uni_d(pend[1]), xs2 = uni_d(pend[2]), xs3 = R;
            double v0 = vL, v1 = uni_d(vals[1]), v2 = uni_d(vals[2]), v3 = vR;
            VLR_WAVE_FENCE();
            int kk = 0;
            double vb = v0;
            if (v1 > vb) { kk = 1; vb = v1; }
            if (v2 > vb) { kk = 2; vb = v2; }
            if (v3 > vb) { kk = 3; vb = v3; }
            if (kk == 0) { R = xs1; vR = v1; }                              // [L, m1]
            else if (kk == 1) { R = xs2; vR = v2; }                         // [L, m2]
            else if (kk == 2) { L = xs1; vL = v1; }                         // [m1, R]
            else { L = xs2; vL = v2; }                                      // [m2, R]
            (void)xs0; (void)xs3;
        }
        if ((((R - L) >= res) && L < R) || !have_mid) {
            mid = uni_d((R + L) / 2.0);
            have_mid = true;
            if (!have_first) { first_mid = mid; have_first = true; }
            double m1 = (mid + L) / 2.0, m2 = (R + mid) / 2.0;
            if (lane < 3) pend[lane] = (lane == 0) ? mid : (lane == 1) ? m1 : m2;
            np = 3;
            phase = RP_ROUND;
        } else {
            // (the abandoned-arm point of the reference is a first-round point again: see range_advance)
            double lo3 = fmax(mid - res * 3.0, lo);
            double hi3 = fmin(mid + res * 3.0, hi);
            double sa = div3(mid - lo3), sb = div3(hi3 - mid);  // itertools_num::linspace step, n = 4
            if (lane < 6) {
                double v;
                if (lane <= 2) v = lin_pt(lo3, sa, (double)lane);
                else v = lin_pt(mid, sb, (double)(lane - 2));
                pend[lane] = v;
            }
            np = 6;
            phase = RP_TAIL;
        }
        VLR_WAVE_FENCE();
    }
    for (int s = 0; s < c.S; ++s)
        if ((dep >> s) & 1) { ndep++; dep_terms += (unsigned)w->nkeep[s]; }
    if (lane == 0) { w->work[0] += (unsigned long long)(tn * ndep); w->work[1] += (unsigned long long)tn * dep_terms; }
    PROF_ADD(c, 4);  // single-chain rounds
    if (haveBest) map_consider(c, bestJ, inner, bestX);
    if (failed) return __builtin_nan("");
    VLR_WAVE_FENCE();
    VLR_SYNC();
    if (log_on(c)) log_table(c, inner, tx, tv, tn);
    if (phase == RP_SIMPSON) {  // bio LogProb::ln_simpsons_integrate_exp (modes/generic.rs:367-385)
        double M = VLR_NEG_INF, S = 0.0;
        for (int i = 1; i < simpson_n - 1; ++i) lse_add(M, S, tv[i] + log((double)(2 + (i % 2) * 2)));
        lse_add(M, S, tv[0]);
        lse_add(M, S, tv[simpson_n - 1]);
        return lse_value(M, S) + log(hi - lo) - log((double)(simpson_n - 1)) - log(3.0);
    }
    double rr_ = integrate_table(tx, tv, tn, c.sx, c.sv, lane);
    PROF_ADD(c, 5);  // single-chain integrate
    return rr_;
}

// ------------------------------------------------------------------------------------------------
// Row-parallel innermost chains: up to kRows sibling chains (same integrated sample, different outer operands)
// run concurrently, one per 16-lane DPP row.  Every "uniform" control instruction of the adaptive integrator now
// serves four chains; the row's 16 lanes are split into (point, slice) groups for the pileup products and the
// partial products are combined with in-row DPP permutes (a 16-lane row is exactly one DPP row).
__device__ __forceinline__ double row_max(double v, int site = __builtin_LINE()) {
    v = fmax(v, dpp_f64<0xB1>(v, site)); v = fmax(v, dpp_f64<0x4E>(v, site)); v = fmax(v, dpp_f64<0x141>(v, site)); v = fmax(v, dpp_f64<0x140>(v, site));
    return v;
}
__device__ __forceinline__ double row_sum(double v, int site = __builtin_LINE()) {
    v += dpp_f64<0xB1>(v, site); v += dpp_f64<0x4E>(v, site); v += dpp_f64<0x141>(v, site); v += dpp_f64<0x140>(v, site);
    return v;
}
// row maximum of signed 64-bit keys (every lane of the 16-lane row ends with it)
__device__ __forceinline__ long long row_max_i64(long long k, int site = __builtin_LINE()) {
#define VLR_STEP_(CTRL) { const long long o_ = (long long)(((unsigned long long)(unsigned)dpp_i32<CTRL>((int)((unsigned long long)k >> 32), site) << 32) | (unsigned)dpp_i32<CTRL>((int)k, site)); k = o_ > k ? o_ : k; }
    VLR_STEP_(0xB1) VLR_STEP_(0x4E) VLR_STEP_(0x141) VLR_STEP_(0x140)
#undef VLR_STEP_
    return k;
}
__device__ __forceinline__ int row_or(int v, int site = __builtin_LINE()) {
    v |= dpp_i32<0xB1>(v, site); v |= dpp_i32<0x4E>(v, site); v |= dpp_i32<0x141>(v, site); v |= dpp_i32<0x140>(v, site);
    return v;
}

// ln(m) for a mantissa m in [0.5, 1) (what reduce_terms leaves): the classic argument reduction to f in [sqrt(1/2) - 1,
// sqrt(2) - 1], s = f / (2 + f), ln(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2))) with the degree-14 odd minimax polynomial of
// fdlibm's e_log.c (< 1 ulp); no special cases (zero, negative, infinite, subnormal arguments cannot occur), about half the
// instructions of the general log().  Returns ln(m) as hi part; the caller adds E * ln 2.
__device__ __forceinline__ double ln_mantissa(double m) {
#ifdef VLR_DBG_LIBM_LOG  // diagnosis builds: the library logarithm
    return log(m);
#endif
    const bool small = m < 0.70710678118654752440;
    const double mm = small ? m * 2.0 : m;       // [sqrt(1/2), sqrt(2))
    const double kk = small ? -1.0 : 0.0;
    const double f = mm - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double sq = f * r;
    sq = __builtin_fma(__builtin_fma(-d, sq, f), r, sq);  // s = f / (2 + f), correctly rounded to within an ulp
    const double z = sq * sq, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    // k * ln2 split as in fdlibm (k in {-1, 0})
    return __builtin_fma(kk, 6.93147180369123816490e-01, f - ((hfsq - __builtin_fma(sq, hfsq + R, kk * 1.90821492927058770002e-10))));
}

// value of row lane N on every lane of its 16-lane DPP row (row_newbcast: no LDS round trip)
template <int N>
__device__ __forceinline__ double row_bcast(double v, int site = __builtin_LINE()) {
    VLR_XCHK(XK_ROW, site);
#ifdef VLR_DBG_BCAST_SHFL
    return __shfl(v, ((int)__lane_id() & ~15) | N, 64);
#endif
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Products of three points over a lane's observation slice with the coefficient pairs held in REGISTERS (loaded once per
// chain batch: only alpha changes between the rounds of a chain).  Slot j of row lane k is observation k + 16 j; empty slots
// hold {c, q} = {1, 0}, so their term is exactly 1.  Same association as accum_terms_e's two-term groups (mantissas are
// bit-identical); no renormalisation: every term is in [2^-70, 2] (WaveSt::vfast), 13 of them stay normal.
constexpr int kRegSlots = 13;  // 16 * 13 = 208 observations of the integrated sample
#ifdef VLR_DBG_REGHELD  // diagnosis builds: fewer (or no) coefficient pairs held in registers across a batch
constexpr int kRegHeld = VLR_DBG_REGHELD;
#else
constexpr int kRegHeld = 8;    // slots whose coefficient pairs stay in registers for the whole batch; deeper slots of the 13-slot
                               // variant are re-read from LDS every pass (five b128 reads): holding all 13 pushed the
                               // 3-waves-per-SIMD build 20 VGPRs over its budget and the spills around the batch loop went to HBM
#endif
template <int NS>
__device__ __forceinline__ void reg_products(const double* cc, const double* cq, const double* lcoef, int rl, int D, const double* al, double* P) {
    constexpr int NR = NS < kRegHeld ? NS : kRegHeld;
    // smallest pileup this instantiation is chosen for (run_chain_batch): slots below it are filled on every lane's row
    constexpr int DMIN = NS <= 4 ? 1 : (NS <= 8 ? 65 : 129);
    double xc[NS > NR ? NS - NR : 1], xq[NS > NR ? NS - NR : 1];
#pragma unroll
    for (int j = NR; j < NS; ++j) {  // issued first: the LDS latency hides behind the register-held slots
        xc[j - NR] = 1.0; xq[j - NR] = 0.0;
        if (16 * j < DMIN || 16 * j < D) {  // (wave-uniform: slots no lane fills are not read)
            const int i = rl + 16 * j;
            const double* a = lcoef + 2 * (i < D ? i : 0);
            const double c0 = a[0], q0 = a[1];
            xc[j - NR] = i < D ? c0 : 1.0; xq[j - NR] = i < D ? q0 : 0.0;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) P[t] = 1.0;
    // Slot pairs in a fixed association (products of two terms, then of pairs).  A slot that no lane fills holds {1, 0}: its term
    // is exactly 1 and L0 * 1 == L0, so leaving it (or the whole pair) out gives the same bits.  D is wave-uniform: scalar branches.
#pragma unroll
    for (int j = 0; j < NS; j += 2) {
        const double c0 = j < NR ? cc[j < NR ? j : 0] : xc[j >= NR ? j - NR : 0], q0 = j < NR ? cq[j < NR ? j : 0] : xq[j >= NR ? j - NR : 0];
        const double c1 = (j + 1 < NR) ? cc[(j + 1 < NR) ? j + 1 : 0] : xc[(j + 1 >= NR && j + 1 < NS) ? j + 1 - NR : 0];
        const double q1 = (j + 1 < NR) ? cq[(j + 1 < NR) ? j + 1 : 0] : xq[(j + 1 >= NR && j + 1 < NS) ? j + 1 - NR : 0];
        if (j + 1 < NS && (16 * (j + 1) < DMIN || 16 * (j + 1) < D)) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const double L0 = __builtin_fma(q0, al[t], c0);
                const double L1 = __builtin_fma(q1, al[t], c1);
                P[t] *= L0 * L1;
            }
        } else if (16 * j < DMIN || 16 * j < D) {
#pragma unroll
            for (int t = 0; t < 3; ++t) P[t] *= __builtin_fma(q0, al[t], c0);
        }
    }
}
// The same products with the third coefficient (beta != 0: a VAF of exactly one in the sample or its contaminant — one chain
// in ten): every slot comes from memory (c, q from LDS, e from the scratch row), a plain loop over slot pairs with the
// association of the register variant.  Kept out of the register variant: its 13 extra values set the kernel's VGPR peak.
__device__ __forceinline__ void lds_products_e(const double* lcoef, const double* ecoef, int rl, int D, const double* al, const double* be, double* P) {
#pragma unroll
    for (int t = 0; t < 3; ++t) P[t] = 1.0;
    for (int j = 0; 16 * j < D; j += 2) {
        const int i0 = rl + 16 * j, i1 = i0 + 16;
        const bool v0 = i0 < D, v1 = i1 < D;
        const double* a0 = lcoef + 2 * (v0 ? i0 : 0);
        const double* a1 = lcoef + 2 * (v1 ? i1 : 0);
        const double c0 = v0 ? a0[0] : 1.0, q0 = v0 ? a0[1] : 0.0, c1 = v1 ? a1[0] : 1.0, q1 = v1 ? a1[1] : 0.0;
        const double g0 = ld_e(ecoef + (v0 ? i0 : 0)), g1 = ld_e(ecoef + (v1 ? i1 : 0));
        const double e0 = v0 ? g0 : 0.0, e1 = v1 ? g1 : 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double L0 = __builtin_fma(e0, be[t], __builtin_fma(q0, al[t], c0));
            const double L1 = __builtin_fma(e1, be[t], __builtin_fma(q1, al[t], c1));
            P[t] *= L0 * L1;
        }
    }
}

// ---- pass-based chain machine (utils/adaptive_integration.rs:54-131), one chain per 16-lane DPP row, for chains whose
// integrated sample is the only one that moves with the chain.  A pass evaluates up to three points of every row:
// INIT = {lo, hi}; ROUND = {mid, middle1, middle2}; TAIL = 7 points in three passes; Simpson fall-backs = grid points three at
// a time.  The chain state is replicated in the registers of the row's lanes, every lane derives the points itself, joint
// values travel by row broadcasts: the only LDS traffic of a pass is the append to the visited-point table.  The state
// update is written with selects (rows are in different phases; divergent branches cost more than the few dead operations).
struct RegChain {
    double lo, hi, res, fixed, pr0, pr1, pr2, rho, al_fix, be_fix;
    double *tx, *tv;
    const double *ptab, *ecoef;
    int pidx, istride, simpson_n, cap, inner, rl, off, D;
    bool cls_fast, rowon, has_by;
    int tn;
    bool failed, sawnan;
};
// Keyed passes (KEYED): when every row of the batch has the same finite prior value at all of its points and a finite fixed part
// (uniform-prior universes, no l2fc terms — the BASELINE scenarios), the joint values of one chain differ only by the pileup
// product, so the argmax of a round (adaptive_integration.rs:61-94) can be taken on the products themselves: a pass keeps the
// (exponent, mantissa) pair of every product as one ordered 64-bit key — exponent in the high 16 bits, the top 48 fraction bits of
// the mantissa below — compares keys, and appends them to the visited-point table; the logarithms (ln_mantissa: a quarter of the
// instructions of a pass, useful on 12 of 64 lanes there) are taken once per table entry in the batch epilogue, on all lanes.
__device__ __forceinline__ long long product_key(double Pm, int E) {  // Pm in [1/2, 1)
    const unsigned hi = (unsigned)__double2hiint(Pm), lo = (unsigned)__double2loint(Pm);
    const unsigned khi = ((unsigned)E << 16) | ((hi >> 4) & 0xffffu);
    const unsigned klo = (hi << 28) | (lo >> 4);
    return (long long)(((unsigned long long)khi << 32) | klo);
}
__device__ __forceinline__ double key_ln(long long key) {  // ln of the product a key stands for
    const unsigned khi = (unsigned)((unsigned long long)key >> 32), klo = (unsigned)key;
    const int E = (int)khi >> 16;
    const unsigned hi = 0x3fe00000u | ((khi & 0xffffu) << 4) | (klo >> 28), lo = klo << 4;
    return ln_mantissa(__hiloint2double((int)hi, (int)lo)) + (double)E * kLn2;
}
// (mantissa in [1/2, 1), exponent) of the product a key stands for
__device__ __forceinline__ void key_decode(long long key, double& Pm, int& E) {
    const unsigned khi = (unsigned)((unsigned long long)key >> 32), klo = (unsigned)key;
    E = (int)khi >> 16;
    Pm = __hiloint2double((int)(0x3fe00000u | ((khi & 0xffffu) << 4) | (klo >> 28)), (int)(klo << 4));
}
template <int NS, bool KEYED>
__device__ __forceinline__ void reg_chain_loop(Ctx& c, RegChain& q) {
    const DevPlan& p = *c.plan;
    const int rl0 = q.rl, D = q.D, simpson_n = q.simpson_n;
    const double lo = q.lo, hi = q.hi, res = q.res;
    constexpr int NR = NS < kRegHeld ? NS : kRegHeld;
    const double* lcoef = c.coef + 2 * q.off;
    double cc[NR], cq[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // (slots every lane fills — 16 (j + 1) <= D, wave-uniform — need no masks; one slot at most is partial)
        const int i = rl0 + 16 * j;
        if (16 * (j + 1) <= D) {
            const double* a = c.coef + 2 * (q.off + i);
            cc[j] = a[0]; cq[j] = a[1];
        } else if (16 * j < D) {
            const double* a = c.coef + 2 * (q.off + (i < D ? i : 0));
            const double c0 = a[0], q0 = a[1];
            cc[j] = i < D ? c0 : 1.0; cq[j] = i < D ? q0 : 0.0;
        } else { cc[j] = 1.0; cq[j] = 0.0; }
    }
    // The rows of a batch run the same PHASE at the same time (wave-uniform `up`, scalar branches): Simpson grids first (rare),
    // then INIT, the ROUNDs while any row's bracket is still wider than its resolution (rows that are through wait: the tail has the
    // same length for every row, so the slowest row sets the pass count either way), then the three TAIL passes.  Only the bracket
    // update of a ROUND needs per-row selects.
    enum { UP_SIMPSON = 0, UP_INIT = 1, UP_ROUND = 2, UP_TAIL = 3 };
    bool live = q.rowon;                        // the row has not run out of table space
    const bool simp = simpson_n != 0;           // per row
    bool act = !simp;                           // the row's search goes on (ROUND)
    const unsigned long long any_simp = __ballot(live && simp), any_norm = __ballot(live && !simp);
    int up = any_simp ? UP_SIMPSON : UP_INIT;
    const int kmax = any_simp ? max(max(VLR_RDLANE(simpson_n, 0), VLR_RDLANE(simpson_n, 16)),
                                    max(VLR_RDLANE(simpson_n, 32), VLR_RDLANE(simpson_n, 48))) : 0;
    double sstep = 0.0;
    if (any_simp) sstep = simp ? (hi - lo) / (double)(simpson_n - 1) : 0.0;  // (a division: only where a grid is walked)
    int k = 0, tn = 0;
    bool failed = false, sawnan = false;
    const bool all_fast = __ballot(q.rowon && !q.cls_fast) == 0ull;
#ifdef VLR_DBG_NO_ONES_PASS
    const bool ones_on = false;
#else
    const bool ones_on = ones_any(c) && ones_risk(c, q.inner);
#endif
    const bool cap_safe = p.table_cap < kTableCapMax;  // the host's bound was not clamped (vlr_host.cpp: table capacity)
    double L = lo, R = hi, vL = VLR_NEG_INF, vR = VLR_NEG_INF, mid = lo;
    long long kL = 0, kR = 0;  // KEYED: the bracket ends' product keys
    PROF_ADD(c, 7);  // batch prologue (task fields, coefficient registers)
    for (;;) {
        PROF_ADD(c, 15);
        const int rl = (((VLR_FRESH_MASK >> 4) & 1u) ? fresh_lane(rl0) : (rl0));  // lane masks (rl == 1, rl < nn, ...) are recomputed: one compare each instead of two lane reads of a spilled pair
        double px0, px1, px2;
        int nn;
        bool on;
        if (__builtin_expect(up == UP_ROUND, 1)) {
            on = live && act;
            px0 = (R + L) / 2.0; px1 = (px0 + L) / 2.0; px2 = (R + px0) / 2.0; nn = 3;
            mid = on ? px0 : mid;
        } else if (up == UP_TAIL) {
            // small interval around the optimum (107-131): three points below, three above (the abandoned-arm point of the
            // reference is a first-round point again: see range_advance)
            on = live && !simp;
            if (k == 0) {
                const double lo3 = fmax(mid - res * 3.0, lo);
                const double sa = div3(mid - lo3);  // itertools_num::linspace step, n = 4
                px0 = lin_pt(lo3, sa, 0.0); px1 = lin_pt(lo3, sa, 1.0); px2 = lin_pt(lo3, sa, 2.0);
            } else {
                const double hi3 = fmin(mid + res * 3.0, hi);
                const double sb = div3(hi3 - mid);
                px0 = lin_pt(mid, sb, 1.0); px1 = lin_pt(mid, sb, 2.0); px2 = lin_pt(mid, sb, 3.0);
            }
            nn = 3;
        } else if (up == UP_INIT) {
            on = live && !simp;
            px0 = lo; px1 = hi; px2 = hi; nn = 2;
        } else {  // Simpson grid (modes/generic.rs:367-385): points k, k+1, k+2 of linspace(lo, hi, n) with exact end points
            const int left = simpson_n - k;
            on = live && simp && left > 0;
            nn = left < 3 ? left : 3;
            px0 = k == 0 ? lo : lin_pt(lo, sstep, (double)k);
            px1 = (k + 1 == simpson_n - 1) ? hi : lin_pt(lo, sstep, (double)(k + 1));
            px2 = (k + 2 == simpson_n - 1) ? hi : lin_pt(lo, sstep, (double)(k + 2));
            px2 = nn < 3 ? px1 : px2;
            px1 = nn < 2 ? px0 : px1; px2 = nn < 2 ? px0 : px2;
        }
        if (__builtin_expect(!cap_safe, 0)) {  // (an unclamped capacity covers every chain: 2 + 3 rounds + 6 points, rounds <= log_{4/3}(1/resolution) + 1)
            const bool over = on && (tn + nn > q.cap);
            failed = failed || over;
            live = live && !over;
            on = on && !over;
        }
        TRC(c, 50, up); TRC(c, 51, k); TRC(c, 52, px0); TRC(c, 53, px1); TRC(c, 54, px2); TRC(c, 55, nn); TRC(c, 56, on ? 1 : 0); TRC(c, 57, rl); TRC(c, 58, tn);
        double al[3], be[3], P[3];
        int E[3];
        {
            const double xs[3] = {px0, px1, px2};
            if (q.has_by) {  // (wave-uniform: the integrated sample has a contaminant)
#pragma unroll
                for (int t = 0; t < 3; ++t) al[t] = q.rho * xs[t] + q.al_fix;
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) al[t] = xs[t];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) be[t] = 0.0;
            if (q.ecoef != nullptr) {  // beta only matters where the third coefficients are not all zero
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const double ind = xs[t] == 1.0 ? 1.0 : 0.0;
                    be[t] = q.has_by ? q.rho * ind + q.be_fix : ind;
                }
            }
        }
        VLR_PRIO_PRODUCTS();
        if (q.ecoef != nullptr && __ballot(on && (be[0] != 0.0 || be[1] != 0.0 || be[2] != 0.0)) != 0ull)
            lds_products_e(lcoef, q.ecoef, rl, D, al, be, P);
        else
            reg_products<NS>(cc, cq, lcoef, rl, D, al, P);
        VLR_PRIO_CHAIN();
        TRC(c, 60, P[0]); TRC(c, 61, P[1]); TRC(c, 62, P[2]); TRC(c, 63, al[0]); TRC(c, 64, al[1]); TRC(c, 65, al[2]);
        PROF_ADD(c, 12);  // pass: term products
        // reduction over the 16 lanes of the row, transposed from the first step on: a lane and its neighbour (xor 1) exchange what the
        // OTHER keeps — the even lane goes on with points 0 and 2, the odd one with point 1 (and a copy of 2) —, then the halves of a
        // quad (xor 2) do the same, lane t of every quad ends with the quad's product of point t, and the four quads are combined for
        // that point alone (rotations by 4 and 8 lanes): lane t of the row ends with point t.  Five multiply / exchange groups instead
        // of eight; the products are the same pairs in the same association as a full butterfly (a b = b a), bit for bit.
#ifdef VLR_FULL_BUTTERFLY
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int e;
            P[t] = __builtin_frexp(P[t], &e); E[t] = e;
            P[t] *= dpp_f64<0xB1>(P[t]); E[t] += dpp_i32<0xB1>(E[t]);      // quad_perm [1,0,3,2]
            P[t] *= dpp_f64<0x4E>(P[t]); E[t] += dpp_i32<0x4E>(E[t]);      // quad_perm [2,3,0,1]
        }
        const int tq = rl & 3;
        double Psel = tq == 1 ? P[1] : tq == 2 ? P[2] : P[0];
        int Esel = tq == 1 ? E[1] : tq == 2 ? E[2] : E[0];
#else
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int e;
            P[t] = __builtin_frexp(P[t], &e); E[t] = e;
        }
        const int tq = rl & 3;
        const bool odd = (tq & 1) != 0;
        // neighbours: the even lane hands over its share of point 1 and takes the odd lane's share of point 0
        const double sendA = odd ? P[0] : P[1], ownA = odd ? P[1] : P[0];
        const int sendAE = odd ? E[0] : E[1], ownAE = odd ? E[1] : E[0];
        const double KA = ownA * dpp_f64<0xB1>(sendA);                     // even: point 0, odd: point 1 (quad_perm [1,0,3,2])
        const int KAE = ownAE + dpp_i32<0xB1>(sendAE);
        const double KB = P[2] * dpp_f64<0xB1>(P[2]);                      // point 2 on both
        const int KBE = E[2] + dpp_i32<0xB1>(E[2]);
        // halves of the quad: lane 0 wants the other even lane's point 0, lane 2 its point 2; lanes 1 and 3 both go on with point 1
        const double sendB = tq == 0 ? KB : KA, ownB = tq == 2 ? KB : KA;
        const int sendBE = tq == 0 ? KBE : KAE, ownBE = tq == 2 ? KBE : KAE;
        double Psel = ownB * dpp_f64<0x4E>(sendB);                         // quad_perm [2,3,0,1]
        int Esel = ownBE + dpp_i32<0x4E>(sendBE);
#endif
        Psel *= dpp_f64<0x124>(Psel); Esel += dpp_i32<0x124>(Esel);        // row_ror:4
        Psel *= dpp_f64<0x128>(Psel); Esel += dpp_i32<0x128>(Esel);        // row_ror:8
        {
            int e;
            Psel = __builtin_frexp(Psel, &e);  // product of 16 mantissas >= 2^-16: one renormalisation suffices
            Esel += e;
        }
        TRC(c, 66, Psel); TRC(c, 67, Esel);
        PROF_ADD(c, 13);  // pass: reduction
        const double x = rl == 0 ? px0 : rl == 1 ? px1 : px2;
        const bool owner = on && rl < nn;
        if (__builtin_expect(ones_on, 0) && up != UP_ROUND) {  // (a middle point of a round is never exactly the range end)
            // this row's chain is at the all-ones point at x == 1 when its contaminant sits at 1 as well (al_fix = irho * contaminant VAF);
            // nothing of this is kept in registers across the passes
            double oP; int oE;
            ones_load(c, q.inner, oP, oE);
            const bool o1 = q.rowon && x == 1.0 && (!q.has_by || q.al_fix == p.irho[q.inner]);
            Psel = o1 ? oP : Psel; Esel = o1 ? oE : Esel;
        }
        double joint = 0.0;
        long long key = 0;
        if (KEYED) {
            key = product_key(Psel, Esel);
#ifdef VLR_DBG_UNIFORM_STORES  // diagnosis builds: the table append without a divergent region (lanes that own no point write to a scratch row)
            { double* dx_ = owner ? q.tx + (tn + rl) : c.w->bpend[0] + (c.lane & 31); double* dv_ = owner ? q.tv + (tn + rl) : c.w->bvals[0] + (c.lane & 31);
              *dx_ = x; *dv_ = __longlong_as_double(key); }
#else
            if (owner) { q.tx[tn + rl] = x; q.tv[tn + rl] = __longlong_as_double(key); }
#endif
        } else {
            double lm = ln_mantissa(Psel);
            if (__builtin_expect(ones_on, 0)) lm = ln_product_mantissa(Psel);  // (a direct all-ones product may be exactly zero)
            const double lik = q.fixed + (lm + (double)Esel * kLn2);
            if (__builtin_expect(c.nlfc > 0, 0) && !lfcs_ok(c, q.inner, x)) joint = VLR_NEG_INF;
            else if (__builtin_expect(all_fast, 1)) joint = (x == 0.0 ? q.pr0 : q.pr1) + lik;  // every row inside a uniform-prior universe: class 0 at exactly 0, else 1
            else {
                const int cls = q.cls_fast ? (x == 0.0 ? 0 : 1) : prior_class(p, q.inner, x);
                const double pv = cls == 0 ? q.pr0 : cls == 1 ? q.pr1 : cls == 2 ? q.pr2 : q.ptab[q.pidx + cls * q.istride];
                joint = pv + lik;
            }
            sawnan = sawnan || (owner && joint != joint);
#ifdef VLR_DBG_UNIFORM_STORES
            { double* dx_ = owner ? q.tx + (tn + rl) : c.w->bpend[0] + (c.lane & 31); double* dv_ = owner ? q.tv + (tn + rl) : c.w->bvals[0] + (c.lane & 31);
              *dx_ = x; *dv_ = joint; }
#else
            if (owner) { q.tx[tn + rl] = x; q.tv[tn + rl] = joint; }
#endif
        }
        TRCB(c, 68, key); TRC(c, 69, joint); TRC(c, 70, owner ? 1 : 0); TRC(c, 71, x);
        tn = on ? tn + nn : tn;
        PROF_ADD(c, 14);  // pass: log + prior + store
        if (__builtin_expect(up == UP_ROUND, 1)) {
            // argmax over {left, middle1, middle2, right}, lowest index wins ties (adaptive_integration.rs:70-82): as three
            // compare masks.  0: [L, m1]  1: [L, m2]  2: [m1, R]  3: [m2, R] — the new bracket keeps one end and takes one of the
            // two middles, so the update is two selects for the middle and one per bracket field
            bool keepR, useM2;
            if (KEYED) {
                const long long j1 = __double_as_longlong(row_bcast<1>(__longlong_as_double(key))), j2 = __double_as_longlong(row_bcast<2>(__longlong_as_double(key)));
                const bool c1 = j1 > kL;
                const long long vb1 = c1 ? j1 : kL;
                const bool c2 = j2 > vb1;
                const long long vb2 = c2 ? j2 : vb1;
                const bool c3 = kR > vb2;
                keepR = c3 || c2;                 // kk >= 2: the left end moves
                useM2 = c3 || (!c2 && c1);        // kk odd: the moving end goes to middle2
                const long long mK = useM2 ? j2 : j1;
                kL = (on && keepR) ? mK : kL; kR = (on && !keepR) ? mK : kR;
            } else {
                const double j1 = row_bcast<1>(joint), j2 = row_bcast<2>(joint);
                const bool c1 = j1 > vL;
                const double vb1 = c1 ? j1 : vL;
                const bool c2 = j2 > vb1;
                const double vb2 = c2 ? j2 : vb1;
                const bool c3 = vR > vb2;
                keepR = c3 || c2;
                useM2 = c3 || (!c2 && c1);
                const double mV = useM2 ? j2 : j1;
                vL = (on && keepR) ? mV : vL; vR = (on && !keepR) ? mV : vR;
            }
            const double mX = useM2 ? px2 : px1;
            const bool updL = on && keepR, updR = on && !keepR;
            L = updL ? mX : L; R = updR ? mX : R;
            TRC(c, 72, L); TRC(c, 73, R); TRCB(c, 74, kL); TRCB(c, 75, kR); TRC(c, 76, vL); TRC(c, 77, vR); TRC(c, 78, keepR ? 1 : 0); TRC(c, 79, useM2 ? 1 : 0);
            act = on && ((R - L) >= res) && L < R;
            if (!__ballot(act)) { up = UP_TAIL; k = 0; }
        } else if (up == UP_TAIL) {
            k += 3;
            if (k >= 6) break;
        } else if (up == UP_INIT) {
            if (KEYED) {
                kL = __double_as_longlong(row_bcast<0>(__longlong_as_double(key))); kR = __double_as_longlong(row_bcast<1>(__longlong_as_double(key)));
            } else {
                vL = row_bcast<0>(joint); vR = row_bcast<1>(joint);  // rows that are not on never read them
            }
            up = UP_ROUND;                                        // the first round always happens (middle is None)
        } else {
            k += 3;
            if (k >= kmax) {
                if (!any_norm) break;
                up = UP_INIT; k = 0;
            }
        }
    }
    PROF_ADD(c, 15);  // pass: state update (+ loop control)
    TRC(c, 80, tn);
    q.tn = tn; q.failed = failed; q.sawnan = sawnan;
}


// ---- ranks of a row's visited points by a bitonic network on 32-bit keys (batch epilogue).  The 16 lanes of a DPP row hold
// N = 16 << NB keys, lane rl the sorted positions (rl << NB) .. (rl << NB) + (1 << NB) - 1.  A compare-exchange along a lane bit takes
// the partner's key with ds_swizzle (the LDS crossbar: no VALU slot, no memory) and keeps the smaller or the larger key — one compare,
// one scalar xor with the stage's lane mask, one select; along a register bit it is one compare and two selects.  146 VALU
// instructions for 64 keys, where ranking every key against every other one is 2 x 4 x 57.
template <int XOR>
__device__ __forceinline__ unsigned swz_xor(unsigned v, int site = __builtin_LINE()) { VLR_XCHK(XK_ROW, site); return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (XOR << 10) | 0x1f); }
// lanes that keep the LARGER key in the cross-lane stage (k, j) / whose register pairs are sorted descending in a register stage of merge k
template <int NB>
__device__ constexpr unsigned long long bitonic_keepmax(int k, int j) {
    unsigned long long m = 0;
    for (int L = 0; L < 64; ++L) {
        const int e = (L & 15) << NB;
        const bool keep_min = ((e & j) == 0) == ((e & k) == 0);
        if (!keep_min) m |= 1ull << L;
    }
    return m;
}
template <int NB>
__device__ constexpr unsigned long long bitonic_desc(int k) {
    unsigned long long m = 0;
    for (int L = 0; L < 64; ++L)
        if ((((L & 15) << NB) & k) != 0) m |= 1ull << L;
    return m;
}
__device__ __forceinline__ unsigned bitonic_pick(unsigned a, unsigned b, unsigned long long keepmax) {
    unsigned r;
    asm("v_cmp_lt_u32_e32 vcc, %1, %2\n\ts_xor_b64 vcc, vcc, %3\n\tv_cndmask_b32_e32 %0, %2, %1, vcc" : "=v"(r) : "v"(a), "v"(b), "s"(keepmax) : "vcc");
    return r;
}
__device__ __forceinline__ void bitonic_pair(unsigned& lo, unsigned& hi, unsigned long long desc) {
    unsigned nl, nh;
    asm("v_cmp_lt_u32_e32 vcc, %2, %3\n\ts_xor_b64 vcc, vcc, %4\n\tv_cndmask_b32_e32 %0, %3, %2, vcc\n\tv_cndmask_b32_e32 %1, %2, %3, vcc"
        : "=&v"(nl), "=&v"(nh) : "v"(lo), "v"(hi), "s"(desc) : "vcc");
    lo = nl; hi = nh;
}
template <int NB, int K, int J>
__device__ __forceinline__ void bitonic_stage(unsigned* s) {
    constexpr int R = 1 << NB;
    if constexpr (J >= R) {
        constexpr unsigned long long km = bitonic_keepmax<NB>(K, J);
#pragma unroll
        for (int t = 0; t < R; ++t) s[t] = bitonic_pick(s[t], swz_xor<(J >> NB)>(s[t]), km);
    } else {
#pragma unroll
        for (int t = 0; t < R; ++t) {
            if ((t & J) == 0) {
                // ascending where (e & K) == 0: a lane mask for K >= R, a property of the register pair below that
                constexpr unsigned long long lane_desc = bitonic_desc<NB>(K);
                const unsigned long long desc = (K >= R) ? lane_desc : ((t & K) != 0 ? ~0ull : 0ull);
                bitonic_pair(s[t], s[t | J], desc);
            }
        }
    }
    if constexpr (J > 1) bitonic_stage<NB, K, (J >> 1)>(s);
}
template <int NB, int K>
__device__ __forceinline__ void bitonic_merge_all(unsigned* s) {
    bitonic_stage<NB, K, (K >> 1)>(s);
    if constexpr (K < (16 << NB)) bitonic_merge_all<NB, (K << 1)>(s);
}
// rank[t] of the row's entry rl + 16 t (t < 1 << NB) among the row's n entries by (key of x, table index); rb: 64 bytes of the row's LDS
template <int NB>
__device__ __forceinline__ void bitonic_ranks(const double* xi, int n, int rl, double lo, double scale, unsigned char* rb, int* rank) {
    constexpr int R = 1 << NB;
    unsigned s[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int i = rl + 16 * t;
        const unsigned q26 = (unsigned)((xi[t] - lo) * scale);   // (entries beyond n hold +inf: saturates; they get the key ~0 anyway)
        s[t] = i < n ? ((q26 << 6) | (unsigned)i) : 0xffffffffu;
    }
    bitonic_merge_all<NB, 2>(s);
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int e = (rl << NB) | t;
        if (e < n) rb[s[t] & 63u] = (unsigned char)e;
    }
    VLR_WAVE_FENCE();
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int i = rl + 16 * t;
        rank[t] = i < n ? (int)rb[i] : 0;
    }
}

__device__ __forceinline__ void run_chain_batch(Ctx& c, int rowmask, int inner) {
    rowmask = UNI(rowmask); inner = UNI(inner);
    PROF_ADD(c, 6);  // batch preparation (task setup, fixed-sample likelihoods)
#ifdef VLR_PROFILE
    c.prof[10] += 1;
#endif
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 5) & 1u) ? fresh_lane(c.lane) : (c.lane)), row = lane >> 4, rl = lane & 15;
    const bool rowon = (rowmask >> row) & 1;
    const int cap = c.cap;
    VLR_WAVE_FENCE();
    const ChainTask& T = w->task[rowon ? row : __builtin_ctz(rowmask)];
    const double lo = T.lo, hi = T.hi, res = T.res, fixed0 = T.fixed;
    const RangeV orig{T.ostart, T.oend, T.olex, T.orex};
    const int simpson_n = T.simpson_n, pidx = T.pidx, contained = T.contained;
    const double* tvr = c.tvaf + (rowon ? row : __builtin_ctz(rowmask)) * c.S;
    double* tx = c.rowX + row * cap;
    double* tv = c.rowV + row * cap;
    double* pend = w->bpend[row];
    double* vals = w->bvals[row];
    int dep = 0;
    for (int s = 0; s < c.S; ++s)
        if (s == inner || p.by[s] == inner) dep |= 1 << s;
    const double fixed = fixed0 + kshift_ln(c, dep);  // + the exponents the scaled coefficient pass took out (0 otherwise)
    unsigned dep_terms = 0;  // observation terms per point
    for (int s = 0; s < c.S; ++s)
        if ((dep >> s) & 1) dep_terms += (unsigned)w->nkeep[s];
    const double* ptab = p.prior_table + c.vt * p.table_size;
    const int istride = p.class_stride[inner];
    // prior values of the first classes of the integrated sample, loaded once (a global load per round would sit
    // on the critical path of every round)
    const int ncls = p.n_class[inner];
    const double pr0 = ptab[pidx], pr1 = ncls > 1 ? ptab[pidx + istride] : VLR_NEG_INF, pr2 = ncls > 2 ? ptab[pidx + 2 * istride] : VLR_NEG_INF;
    // prior class of the integrated sample: if one Range spectrum of a uniform-prior universe covers [lo, hi],
    // every point of the chain is inside the universe (class 1, or 0 at exactly 0) — no per-point spectrum walk
    bool cls_fast = false;
    if (p.prior_kind[inner] == PK_UNIFORM)
        for (int u = p.uni_off[inner]; u < p.uni_off[inner + 1]; ++u) {
            const DevSpectrum sp = ld_spec(p.universe + u);
            if (sp.kind == 1) {
                RangeV ur{sp.start, sp.end, sp.lex, sp.rex};
                cls_fast = cls_fast || (range_contains(ur, lo) && range_contains(ur, hi));
            }
        }

    int np, phase, tn = 0;
    bool done = !rowon;
    double L = lo, R = hi, vL = VLR_NEG_INF, vR = VLR_NEG_INF, mid = lo, first_mid = lo;
    bool have_first = false, have_mid = false, failed = false, sawnan = false;
    const int D_in = UNI(w->nkeep[inner]);
    // register-resident runner: the integrated sample is the only one whose likelihood moves with the chain (no sample is
    // contaminated by it), its pileup fits the register slots and its terms need no renormalisation
#ifdef VLR_DBG_NO_REGRUN  // diagnosis builds: every batch through the round-1 loop (coefficients from LDS, per-term renormalisation)
    const bool regrun = false && dep == (1 << inner);
#else
    const bool regrun = dep == (1 << inner) && D_in <= 16 * kRegSlots && ((UNI(w->vfast) >> inner) & 1);
#endif
    TRC(c, 30, lo); TRC(c, 31, hi); TRC(c, 32, res); TRC(c, 33, fixed); TRC(c, 34, simpson_n); TRC(c, 35, pidx); TRC(c, 36, rowon ? 1 : 0);
    TRC(c, 37, regrun ? 1 : 0); TRC(c, 38, D_in); TRC(c, 39, pr0); TRC(c, 40, pr1);
    // keyed passes (see reg_chain_loop): every point of every row has the same finite prior value and a finite fixed part
#ifdef VLR_DBG_NO_KEYED  // diagnosis builds: every pass takes the logarithm itself
    const bool keyed = false && c.nlfc == 0 &&
#else
    const bool keyed = regrun && c.nlfc == 0 && !(ones_any(c) && ones_risk(c, inner)) &&
#endif  // (the exponent of a direct all-ones product is not bounded by the key's 16 bits)
                       __ballot(rowon && !(cls_fast && (pr0 == pr1 || lo != 0.0) && fabs(pr1) < __builtin_huge_val() && fabs(fixed) < __builtin_huge_val())) == 0ull;
    TRC(c, 41, keyed ? 1 : 0); TRC(c, 42, cls_fast ? 1 : 0);
    if (__builtin_expect(regrun, 1)) {
        RegChain rc;
        rc.lo = lo; rc.hi = hi; rc.res = res; rc.fixed = fixed; rc.pr0 = pr0; rc.pr1 = pr1; rc.pr2 = pr2;
        rc.tx = tx; rc.tv = tv; rc.ptab = ptab; rc.pidx = pidx; rc.istride = istride; rc.simpson_n = simpson_n;
        rc.cap = cap; rc.cls_fast = cls_fast; rc.rowon = rowon; rc.inner = inner; rc.rl = rl;
        rc.off = UNI(w->soff[inner]); rc.D = D_in;
        const int byi = p.by[inner];
        rc.has_by = byi >= 0;
        rc.rho = p.rho[inner];
        const double bvaf = byi >= 0 ? tvr[byi] : 0.0;  // contaminant VAF: fixed along the chain
        rc.al_fix = p.irho[inner] * bvaf; rc.be_fix = p.irho[inner] * (bvaf == 1.0 ? 1.0 : 0.0);
        rc.ecoef = ecoef_of(c, inner, rc.off);
        if (keyed) {
            if (D_in <= 64) reg_chain_loop<4, true>(c, rc);
            else if (D_in <= 128) reg_chain_loop<8, true>(c, rc);
            else reg_chain_loop<kRegSlots, true>(c, rc);
        } else {
            if (D_in <= 64) reg_chain_loop<4, false>(c, rc);
            else if (D_in <= 128) reg_chain_loop<8, false>(c, rc);
            else reg_chain_loop<kRegSlots, false>(c, rc);
        }
        tn = rc.tn; failed = rc.failed; sawnan = rc.sawnan;
        phase = simpson_n ? RP_SIMPSON : RP_TAIL;
        np = 0;
    } else {
    if (simpson_n) {
        double step = (hi - lo) / (double)(simpson_n - 1);
        if (rl < simpson_n) pend[rl] = (rl == 0) ? lo : (rl == simpson_n - 1) ? hi : lin_pt(lo, step, (double)rl);
        np = simpson_n;
        phase = RP_SIMPSON;
    } else {
        if (rl < 2) pend[rl] = rl ? hi : lo;
        np = 2;
        phase = RP_INIT;
    }
    VLR_WAVE_FENCE();

    while (__ballot(!done)) {
        const bool act = !done;
#ifdef VLR_PROFILE
        c.prof[10] += 1ull << 32;                                                     // rounds (high word)
        c.prof[11] += (unsigned long long)(popc64(__ballot(!done && rl == 0))) << 32;  // active rows (high word)
#endif
        if (act && tn + np > cap) { failed = true; done = true; }
        const bool go = act && !failed;
        // row lane j < np owns point j of its chain; every lane multiplies its observation slice (terms rl, rl+16,
        // ...) for up to four points per pass.  Rows in different phases have different np: the pass loop runs to
        // the largest, rows without points left compute on repeated points and discard the result.
        PROF_ADD(c, 15);  // round: advance + loop control (previous iteration)
        const int npg = go ? np : 0;
        const int npmax = max(max(VLR_RDLANE(npg, 0), VLR_RDLANE(npg, 16)),
                              max(VLR_RDLANE(npg, 32), VLR_RDLANE(npg, 48)));
        const double x = pend[rl < np ? rl : np - 1];
        double Psel = 1.0;
        int Esel = 0;
        for (int p0 = 0; p0 < npmax; p0 += kPass) {
            const int cnt = (npmax - p0) < kPass ? (npmax - p0) : kPass;
            double xs[kPass], P[kPass];
            int E[kPass];
#pragma unroll
            for (int j = 0; j < kPass; ++j) { xs[j] = pend[(p0 + j) < np ? (p0 + j) : np - 1]; P[j] = 1.0; E[j] = 0; }
            int dm = dep;
            while (dm) {
                const int s = __builtin_ctz(dm);
                dm &= dm - 1;
                const int by = p.by[s];
                const double va = tvr[s], vb = by >= 0 ? tvr[by] : 0.0;
                double al[kPass], be[kPass];
#pragma unroll
                for (int j = 0; j < kPass; ++j) {
                    const double a = (s == inner) ? xs[j] : va;
                    const double b = by >= 0 ? ((by == inner) ? xs[j] : vb) : 0.0;
                    alpha_beta(p, s, a, b, al[j], be[j]);
                }
                bool one_j[kPass];
                bool any_one = false;
                if (__builtin_expect(ones_any(c), 0) && ones_risk(c, s)) {
#pragma unroll
                    for (int j = 0; j < kPass; ++j) {
                        const double a = (s == inner) ? xs[j] : va;
                        const double b = by >= 0 ? ((by == inner) ? xs[j] : vb) : 0.0;
                        one_j[j] = is_all_ones(p, s, a, b);
                        any_one = any_one || one_j[j];
                    }
                }
                if (__builtin_expect(__ballot(any_one) != 0ull, 0)) {
                    double Pt[kPass];
                    int Et[kPass];
#pragma unroll
                    for (int j = 0; j < kPass; ++j) { Pt[j] = 1.0; Et[j] = 0; }
                    accum_terms_n<16>(cnt, c.coef + 2 * UNI(w->soff[s]), ecoef_of(c, s, UNI(w->soff[s])), UNI(w->nkeep[s]), rl, (UNI(w->fastok) >> s) & 1, al, be, Pt, Et);
#pragma unroll
                    for (int j = 0; j < kPass; ++j) {
                        if (one_j[j]) { Pt[j] = 1.0; Et[j] = 0; ones_inject(c, s, rl, Pt[j], Et[j]); }
                        P[j] *= Pt[j]; E[j] += Et[j];
                        int e2;
                        P[j] = __builtin_frexp(P[j], &e2); E[j] += e2;
                    }
                } else
                accum_terms_n<16>(cnt, c.coef + 2 * UNI(w->soff[s]), ecoef_of(c, s, UNI(w->soff[s])), UNI(w->nkeep[s]), rl, (UNI(w->fastok) >> s) & 1, al, be, P, E);
            }
            PROF_ADD(c, 12);  // round: term products
            reduce_terms_n<16>(cnt, P, E);
            const int jr = rl - p0;
            double Pm = P[0];
            int Em = E[0];
#pragma unroll
            for (int j = 1; j < kPass; ++j) { Pm = (jr == j) ? P[j] : Pm; Em = (jr == j) ? E[j] : Em; }
            const bool mine = jr >= 0 && jr < kPass;
            Psel = mine ? Pm : Psel;
            Esel = mine ? Em : Esel;
        }
        PROF_ADD(c, 13);  // round: reduction
        const double lik = fixed + (log(Psel) + (double)Esel * kLn2);
        double joint;
        if (c.nlfc > 0 && !lfcs_ok(c, inner, x)) joint = VLR_NEG_INF;
        else {
            int cls = cls_fast ? (x == 0.0 ? 0 : 1) : prior_class(p, inner, x);
            double pv = cls == 0 ? pr0 : cls == 1 ? pr1 : cls == 2 ? pr2 : ptab[pidx + cls * istride];
            joint = pv + lik;
        }
        const bool owner = go && rl < np;
        if (owner && joint != joint) sawnan = true;
        if (owner) { tx[tn + rl] = x; tv[tn + rl] = joint; vals[rl] = joint; }
        PROF_ADD(c, 14);  // round: log + prior + store
        VLR_WAVE_FENCE();
        if (go) {
            tn += np;
            if (phase == RP_SIMPSON || phase == RP_TAIL) done = true;
            else {
                if (phase == RP_INIT) { L = lo; R = hi; vL = vals[0]; vR = vals[1]; }
                else {  // argmax over {left, middle1, middle2, right}; lowest index wins ties
                    double xs1 = pend[1], xs2 = pend[2];
                    double v1 = vals[1], v2 = vals[2];
                    int kk = 0;
                    double vb = vL;
                    if (v1 > vb) { kk = 1; vb = v1; }
                    if (v2 > vb) { kk = 2; vb = v2; }
                    if (vR > vb) { kk = 3; }
                    if (kk == 0) { R = xs1; vR = v1; }
                    else if (kk == 1) { R = xs2; vR = v2; }
                    else if (kk == 2) { L = xs1; vL = v1; }
                    else { L = xs2; vL = v2; }
                }
                VLR_WAVE_FENCE();
                if ((((R - L) >= res) && L < R) || !have_mid) {
                    mid = (R + L) / 2.0;
                    have_mid = true;
                    if (!have_first) { first_mid = mid; have_first = true; }
                    double m1 = (mid + L) / 2.0, m2 = (R + mid) / 2.0;
                    if (rl < 3) pend[rl] = (rl == 0) ? mid : (rl == 1) ? m1 : m2;
                    np = 3;
                    phase = RP_ROUND;
                } else {
                    // (the abandoned-arm point of the reference is a first-round point again: see range_advance)
                    double lo3 = fmax(mid - res * 3.0, lo);
                    double hi3 = fmin(mid + res * 3.0, hi);
                    double sa = div3(mid - lo3), sb = div3(hi3 - mid);
                    if (rl < 6) {
                        double v;
                        if (rl <= 2) v = lin_pt(lo3, sa, (double)rl);
                        else v = lin_pt(mid, sb, (double)(rl - 2));
                        pend[rl] = v;
                    }
                    np = 6;
                    phase = RP_TAIL;
                }
            }
        }
        VLR_WAVE_FENCE();
    }
    }  // !regrun
    PROF_ADD(c, 15);
    if (__builtin_expect(__ballot(failed) != 0ull, 0)) c.status |= VLR_LOCUS_TABLE_FULL;
    if (__builtin_expect(__ballot(sawnan) != 0ull, 0)) c.status |= VLR_LOCUS_NAN;
    if (rowon && rl == 0) {  // work counters
        atomicAdd(&w->work[0], (unsigned long long)tn);
        atomicAdd(&w->work[1], (unsigned long long)tn * dep_terms);
    }
    VLR_WAVE_FENCE();

    // ---- epilogue, row-parallel: MAP candidate of the chain and the integral over the visited points.
    // Trapezoid over the sorted grid (LogProb::ln_trapezoidal_integrate_grid_exp, utils/adaptive_integration.rs:133-140)
    // in the linear domain relative to the row maximum M, regrouped per grid point:
    //   sum_seg (e_k + e_{k+1}) (x_{k+1} - x_k)/2  =  sum_k e_k (x_{k+1} - x_{k-1})/2   (one-sided at the ends),
    // so every entry only needs the x of its predecessor and successor in (x, index) order: every lane ranks its (up to
    // four) entries against the row's table — one 64-bit compare and one add-with-carry per pair —, the table is rewritten
    // in rank order in place and the neighbours are read back.  Duplicate x (HashMap key collisions in the reference) are
    // neighbours at distance zero.
    const int n = (rowon && !failed) ? tn : 0;
    const int nmax = max(max(VLR_RDLANE(n, 0), VLR_RDLANE(n, 16)),
                         max(VLR_RDLANE(n, 32), VLR_RDLANE(n, 48)));
    const int TT = (nmax + 15) >> 4;  // entries per lane (uniform), <= 4 because the row path requires cap <= 64
    double bJ = VLR_NEG_INF, bX = 0.0;
    int bHave = 0;
    bool anynan = false, anyout = false;  // anyout: a visited point lies outside the leaf's own range (excluded range end)
    double rint_ = VLR_NEG_INF;
    {
        double xi[4], vi[4];
        int rank[4];
        const bool srt = phase != RP_SIMPSON;  // per row: trapezoid over the sorted visited points (Simpson grids are in order)
        const bool any_simpson = __ballot(rowon && !srt) != 0ull;
        // Keyed batches (round 6): the table holds the product keys of the passes, and everything the epilogue computes is a function of
        // the PRODUCTS — all entries of a row share prior and fixed part, so the arg-best is the largest key (ties: smallest x, as before:
        // equal products are equal joints), the row maximum M is prior + fixed + ln(largest product), and e^(v - M) is the RATIO of two
        // products, (P / P_max) 2^(E - E_max): one reciprocal per row instead of a logarithm and an exponential per entry.  The joint
        // VALUES of the entries are only formed where somebody reads them: the AFD log, the replay pass, and chains whose points are
        // candidates of other groups / lie outside their own range (scan_chain_candidates reads the row table).
        constexpr long long kKeyNone = (long long)0x8000000000000000ull;
        long long kk[4];
        long long bK = kKeyNone;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xi[t] = __builtin_huge_val(); vi[t] = VLR_NEG_INF;
            kk[t] = kKeyNone;
            rank[t] = 0;
            if (t < TT) {
                const int i = rl + 16 * t;
                const bool on = i < n;
                const int ic = on ? i : 0;
                const double xr = tx[ic];
                const double vr = tv[ic];
                xi[t] = on ? xr : __builtin_huge_val();
                const bool inlo = (orig.start < xi[t]) | ((orig.lex == 0) & (orig.start == xi[t]));
                const bool inhi = (orig.end > xi[t]) | ((orig.rex == 0) & (orig.end == xi[t]));
                anyout = anyout | (on & !(inlo & inhi));
                if (keyed) {  // (wave-uniform) the pass stored the product's key; products of keyed passes are finite and positive
                    kk[t] = on ? __double_as_longlong(vr) : kKeyNone;
                    const bool cand = on & (contained != 0) & inlo & inhi;
                    const bool take = cand & ((bHave == 0) | (kk[t] > bK) | ((kk[t] == bK) & (xi[t] < bX)));
                    bK = take ? kk[t] : bK; bX = take ? xi[t] : bX; bHave = take ? 1 : bHave;
                } else {
                    vi[t] = on ? vr : VLR_NEG_INF;
                    const bool cand = on & (contained != 0) & inlo & inhi & (vi[t] == vi[t]);
                    const bool take = cand & ((bHave == 0) | (vi[t] > bJ) | ((vi[t] == bJ) & (xi[t] < bX)));
                    bJ = take ? vi[t] : bJ; bX = take ? xi[t] : bX; bHave = take ? 1 : bHave;
                    anynan = anynan | (vi[t] != vi[t]);
                }
            }
        }
        // who reads the joint values of this batch's entries (wave-uniform)
        const bool need_vals = !keyed || log_on(c) || c.replay != 0 || __ballot(rowon && (T.alive != 0 || contained == 0 || anyout)) != 0ull;
        if (keyed && need_vals) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < TT) vi[t] = (kk[t] != kKeyNone) ? (xi[t] == 0.0 ? pr0 : pr1) + (fixed + key_ln(kk[t])) : VLR_NEG_INF;
        }
        TRC(c, 81, n); TRC(c, 82, nmax); TRC(c, 83, xi[0]); TRC(c, 84, vi[0]); TRC(c, 85, xi[1]); TRC(c, 86, vi[1]); TRC(c, 87, bJ); TRC(c, 88, bX);
        // AFD log: the four row tables as they stand (any order), one record per chain (no l2fc terms on batched chains)
        if (log_on(c)) {
            const int hsz = 1 + c.S;
            int at_row = -1;
#pragma unroll
            for (int r4 = 0; r4 < kRows; ++r4) {
                const int nr = VLR_RDLANE(n, 16 * r4);
                if (nr > 0 && log_on(c)) {
                    const int a = log_reserve(c, hsz + 2 * nr);
                    at_row = (row == r4) ? a : at_row;
                }
            }
            if (at_row >= 0 && n > 0 && c.lg_pos >= 0) {
                const int at = at_row;
                const ChainTask& Tl = w->task[row];
                if (rl == 0) c.lg[at] = __longlong_as_double(log_header(1, n, inner, Tl.disc & ~(1 << inner), Tl.group, 0));
                if (rl < c.S) c.lg[at + 1 + rl] = tvr[rl];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = rl + 16 * t;
                    if (t < TT && i < n) { c.lg[at + hsz + i] = xi[t]; c.lg[at + hsz + n + i] = vi[t]; }
                }
            }
        }
        // rank of every entry among its row's entries, by (x, table index): revisited points (equal x: HashMap key collisions in the
        // reference) get distinct, adjacent ranks.  First choice: a bitonic network on 32-bit keys — 26 bits of (x - lo) / (hi - lo) above
        // the index — whose order is CHECKED against the x values once the table is rewritten; two distinct points closer than 2^-26 of
        // the range fail the check and the exact ranking takes over: one 64-bit compare and one add-with-carry per (entry, q) against
        // keys parked in the value table (x and value of every entry are in registers by now).
        if (__ballot(srt && n > 0)) {
#ifdef VLR_NO_BITONIC
            bool use32 = false;
#else
            bool use32 = true;
            {
                const double span = hi - lo;
                const double scale = span > 0.0 ? 67108863.0 / span : 0.0;   // (2^26 - 1) / (hi - lo)
                VLR_WAVE_FENCE();
                if (nmax > 32) bitonic_ranks<2>(xi, n, rl, lo, scale, (unsigned char*)tv, rank);
                else if (nmax > 16) bitonic_ranks<1>(xi, n, rl, lo, scale, (unsigned char*)tv, rank);
                else bitonic_ranks<0>(xi, n, rl, lo, scale, (unsigned char*)tv, rank);
            }
#endif
            for (;;) {
                if (__builtin_expect(!use32, 0)) {
                    // sort key: the bit pattern of a non-negative double orders like the number; the low six bits carry the table index.
                    // (Distinct points closer than 64 ulp may swap: a segment of width ~1e-16 changes sign.)
                    unsigned long long key[4];
                    unsigned long long* kv = (unsigned long long*)tv;
                    VLR_WAVE_FENCE();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = rl + 16 * t;
                        key[t] = i < n ? ((((unsigned long long)__double_as_longlong(xi[t])) & ~63ull) | (unsigned long long)i) : ~0ull;
                        rank[t] = 0;
                        if (t < TT && i < cap) kv[i] = key[t];  // ~0 beyond the row's entries
                    }
                    VLR_WAVE_FENCE();
                    int q0 = 0;
                    for (; q0 + 4 <= nmax; q0 += 4) {  // four table reads in flight
                        unsigned long long kq[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) kq[j] = kv[q0 + j];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (t < TT) rank[t] += (kq[j] < key[t]) ? 1 : 0;
                    }
                    for (; q0 < nmax; ++q0) {
                        const unsigned long long kq = kv[q0];
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (t < TT) rank[t] += (kq < key[t]) ? 1 : 0;
                    }
                }
                VLR_WAVE_FENCE();
                // scatter into sorted order, in place (every entry is in registers); rows that are not sorted (Simpson grids) get their
                // values back
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < TT) {
                        const int i = rl + 16 * t;
                        if (i < n) {
                            if (srt) { tx[rank[t]] = xi[t]; if (need_vals) tv[rank[t]] = vi[t]; }
                            else if (need_vals) tv[i] = vi[t];
                        }
                    }
                }
                VLR_WAVE_FENCE();
                if (!use32) break;
                // the order the 32-bit keys gave, checked on the points themselves: no entry may lie below its predecessor
                bool bad = false;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < TT) {
                        const int i = rl + 16 * t;
                        const int rk = rank[t];
                        const double pred = tx[rk > 0 ? rk - 1 : 0];
                        bad = bad || (srt && i < n && rk > 0 && pred > xi[t]);
                    }
                }
                if (__builtin_expect(__ballot(bad) == 0ull, 1)) break;
                use32 = false;
            }
        } else if (keyed && need_vals) {  // no row is sorted (Simpson grids only): the tables still hold keys
            VLR_WAVE_FENCE();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = rl + 16 * t;
                if (t < TT && i < n) tv[i] = vi[t];
            }
            VLR_WAVE_FENCE();
        }
        // ---- row arg-best of (bJ desc, bX asc) for the MAP candidate: the row maximum of the lanes' best joints (a lane
        // without a candidate counts as -inf; whether the row has one at all travels apart: a candidate may be worth -inf itself),
        // then the smallest x among the lanes that hold it
        double M;
        double rP = 0.0;   // keyed: 1 / mantissa of the row's largest product, and its exponent
        int mE = 0;
        if (keyed) {
            long long m4K = kk[0];
#pragma unroll
            for (int t = 1; t < 4; ++t) m4K = kk[t] > m4K ? kk[t] : m4K;
            const long long rowKall = row_max_i64(m4K);
            const long long candK = bHave ? bK : kKeyNone;
            // (a lane whose best candidate is its largest entry — every lane of every row unless points lie outside their own range —
            //  makes the second reduction the first)
            const long long rowK = (__ballot(candK != m4K) == 0ull) ? rowKall : row_max_i64(candK);
            const int rowHave = row_or(bHave);
            double xs_ = (bHave != 0 && bK == rowK) ? bX : __builtin_huge_val();
            xs_ = fmin(xs_, dpp_f64<0xB1>(xs_)); xs_ = fmin(xs_, dpp_f64<0x4E>(xs_)); xs_ = fmin(xs_, dpp_f64<0x141>(xs_)); xs_ = fmin(xs_, dpp_f64<0x140>(xs_));
            double mP;
            key_decode(rowKall, mP, mE);
            M = (n > 0) ? pr1 + (fixed + (ln_mantissa(mP) + (double)mE * kLn2)) : VLR_NEG_INF;
            bJ = M;
            if (__ballot(rowHave != 0 && rowK != rowKall) != 0ull) bJ = (rowK == rowKall) ? M : pr1 + (fixed + key_ln(rowK));
            bX = xs_; bHave = rowHave;
            rP = __builtin_amdgcn_rcp(mP);
            rP = __builtin_fma(__builtin_fma(-mP, rP, 1.0), rP, rP);
            rP = __builtin_fma(__builtin_fma(-mP, rP, 1.0), rP, rP);
        } else {
            const int rowHave = row_or(bHave);
            const double rowJ = row_max(bHave ? bJ : VLR_NEG_INF);
            double xs_ = (bHave != 0 && bJ == rowJ) ? bX : __builtin_huge_val();
            xs_ = fmin(xs_, dpp_f64<0xB1>(xs_)); xs_ = fmin(xs_, dpp_f64<0x4E>(xs_)); xs_ = fmin(xs_, dpp_f64<0x141>(xs_)); xs_ = fmin(xs_, dpp_f64<0x140>(xs_));
            bJ = rowJ; bX = xs_; bHave = rowHave;
            double m4 = VLR_NEG_INF;
#pragma unroll
            for (int t = 0; t < 4; ++t) m4 = fmax(m4, (vi[t] == vi[t]) ? vi[t] : VLR_NEG_INF);
            M = row_max(m4);
        }
        double ssum = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < TT) {
                const int i = rl + 16 * t;
                const bool on = i < n;
                double ev;
                if (keyed) {  // e^(v - M) as the ratio of the two products (entries beyond the row's table: far below the smallest double)
                    double eP; int eE;
                    key_decode(kk[t], eP, eE);
                    ev = __builtin_ldexp(eP * rP, eE - mE);
                } else {
                    const bool zero = (vi[t] == VLR_NEG_INF) | (M == VLR_NEG_INF) | (vi[t] != vi[t]);
                    ev = zero ? 0.0 : exp(vi[t] - M);
                }
                double wgt;
                {
                    // sum_seg (e_k + e_{k+1}) (x_{k+1} - x_k)/2 = sum_k e_k (x_{k+1} - x_{k-1})/2, one-sided at the ends
                    const int rk = on ? rank[t] : 0;
                    const double pred = tx[rk > 0 ? rk - 1 : 0], succ = tx[(rk + 1 < n) ? rk + 1 : rk];
                    const double lo2 = (rk > 0) ? pred : xi[t];
                    const double hi2 = (rk + 1 < n) ? succ : xi[t];
                    wgt = (hi2 - lo2) / 2.0;
                }
                if (any_simpson) wgt = srt ? wgt : ((i == 0 || i == n - 1) ? 1.0 : (double)(2 + (i % 2) * 2));  // Simpson grids (rare)
                ssum += on ? ev * wgt : 0.0;
            }
        }
        TRC(c, 89, rank[0]); TRC(c, 90, rank[1]); TRC(c, 91, M); TRC(c, 92, ssum);
        ssum = row_sum(ssum);
        {   // ln of the positive sum: exponent apart, the mantissa through the short logarithm of the passes (ssum == 0: -inf)
            int es;
            const double ms = __builtin_frexp(ssum, &es);
            const double lns = ssum > 0.0 ? ln_mantissa(ms) + (double)es * kLn2 : VLR_NEG_INF;
            rint_ = (M == VLR_NEG_INF) ? VLR_NEG_INF : M + lns;
        }
    }
    TRC(c, 93, rint_); TRC(c, 94, bJ); TRC(c, 95, bX); TRC(c, 96, bHave);
    int nanrow = row_or(anynan ? 1 : 0);
    const int rowout = row_or(anyout ? 1 : 0);
    double r = rint_;
    if (phase == RP_SIMPSON) r = r + log(hi - lo) - log((double)(simpson_n - 1)) - log(3.0);
    if (nanrow || failed) r = __builtin_nan("");
    if (rowon && rl == 0) {
        ChainTask& To = w->task[row];
        To.result = r; To.bestJ = bJ; To.bestX = bX; To.haveBest = bHave | (rowout ? 2 : 0); To.n = n;
    }
    VLR_WAVE_FENCE();
    VLR_SYNC();
    PROF_ADD(c, 8);  // batch epilogue (MAP scan + integrate)
}

// MAP candidates of one finished row-parallel chain for OTHER groups / containment via another path (rare): the lanes
// test all visited points at once; only points that need the full contains walk are visited one by one, in table order
__device__ inline void scan_chain_candidates(Ctx& c, const double* rx, const double* rvv, int nq, const RangeV& io, int contained, alive_t alive,
                                             int s_in) {
    for (int q0 = 0; q0 < nq; q0 += 64) {
        const int q = q0 + c.lane;
        const bool on = q < nq;
        const double xq = rx[on ? q : 0];
        const bool own = (contained != 0) & range_contains(io, xq);
        const alive_t al = alive ? alive_update(c, alive, s_in, xq) : 0;
        unsigned long long need = __ballot(on & (!own | (al != 0)));
        while (need) {
            const int qq = q0 + __builtin_ctzll(need);
            need &= need - 1;
            const double xqq = uni_d(rx[qq]);
            const alive_t alq = alive ? UNI_A(alive_update(c, alive, s_in, xqq)) : 0;
            map_all(c, uni_d(rvv[qq]), s_in, xqq, false, alq);
        }
    }
}

// LikelihoodOperands::lfc_bounds (modes/generic.rs:148-174)
__device__ inline bool ops_lfc_bounds(const Ctx& c, int sample, RangeV& out) {
    bool have = false;
    RangeV acc = range_empty();
    for (int i = 0; i < c.nlfc; ++i) {
        int sa = c.w->lfc_a[i], sb = c.w->lfc_b[i];
        int cmp = c.w->lfc_cmp[i];
        double val = c.w->lfc_val[i];
        bool got = false;
        RangeV b = range_empty();
        if (sa == sample) {
            if (c.present & (1 << sb)) { lfc_invert(cmp, val); b = lfc_bounds_of(cmp, val, c.w->ops_vaf[sb]); got = true; }
        } else if (sb == sample) {
            if (c.present & (1 << sa)) { b = lfc_bounds_of(cmp, val, c.w->ops_vaf[sa]); got = true; }
        }
        if (got) {
            acc = have ? range_intersect(acc, b) : b;
            have = true;
        }
    }
    out = acc;
    return have;
}

// replay: record the operands of one finished row-parallel chain (row i); ops_vaf must hold the chain's outer operands
__device__ inline void afd_emit_row(Ctx& c, int i, int s_in, int nq) {
    WaveSt* w = c.w;
    const double* rx = c.rowX + i * c.cap;
    const double* rvv = c.rowV + i * c.cap;
    int mism = 0;
    for (int s = 0; s < c.S; ++s)
        if (s != s_in && !(w->ops_vaf[s] == c.mapv[s] && (((c.disc >> s) & 1) == ((c.mapDisc >> s) & 1)))) mism++;
    if (mism == 0) {
        for (int q = 0; q < nq; ++q) { double xq = uni_d(rx[q]); if (!table_has(rx, q, xq, c.lane)) afd_consider(c, uni_d(rvv[q]), s_in, xq); }
    } else if (mism == 1) {
        for (int q = 0; q < nq; ++q) {
            double xq = uni_d(rx[q]);
            if (xq == c.mapv[s_in] && !table_has(rx, q, xq, c.lane)) afd_consider(c, uni_d(rvv[q]), s_in, xq);
        }
    }
}

// Outer Range frame whose single child is a leaf Range node (the nested `somatic_normal` shape): evaluate the
// inner chains of ALL pending outer points together, kRows at a time (run_chain_batch), instead of descending
// once per point.  Semantics identical to the sequential walk (modes/generic.rs:331-395 for the child node).
// The work is split in three steps around run_chain_batch, which the kernel's event loop runs on behalf of the
// walk (bo_begin -> [bo_setup -> run_chain_batch -> bo_deliver]*): inlining the batch runner inside the walk kept
// ~100 more VGPRs alive across it and made the compiler spill in its rounds.
__device__ __forceinline__ void bo_begin(Ctx& c, RangeSt& r, int chn, alive_t alive_in) {
    PROF_ADD(c, 3);
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const DevNode ch = ld_node(p.nodes + chn);
    const int s_in = ch.sample, s_out = UNI(r.sample), S = c.S;
    const int n_obs = UNI(w->nkeep[s_in]);
    const bool clear_ref = n_obs > 10 && UNI((int)w->all_posref[s_in]);
    const RangeV vr{ch.vafs.start, ch.vafs.end, ch.vafs.lex, ch.vafs.rex};
    const bool dead = clear_ref && vr.start > 0.0;  // generic.rs:342-347
    const double res = p.resolution[s_in];
    const double lo = uni_d(observable_min(vr, n_obs)), hi = uni_d(observable_max(vr, n_obs));
    const int simpson = ((hi - lo) < res) ? 3 : (n_obs < 5 ? 11 : 0);
    // samples whose likelihood is fixed during an inner chain: constant over the outer points, or varying with them
    double fixed_const = 0.0;
    int vary = 0;
    for (int s = 0; s < S; ++s) {
        int by = p.by[s];
        if (s == s_in || by == s_in) continue;
        if (s == s_out || by == s_out) vary |= 1 << s;
        else fixed_const += sample_lik(c, s, w->ops_vaf[s], by >= 0 ? w->ops_vaf[by] : 0.0);
    }
    fixed_const = uni_d(fixed_const);
    // cross-event MAP candidates: groups whose spectra for the outer sample miss the whole outer range need no per-point test
    const alive_t alive0 = alive_restrict(c, alive_in, s_out, uni_d(r.lo), uni_d(r.hi));
    VLR_SYNC();
    if (c.lane == 0) {
        BatchOuter& B = w->bo;
        B.alive0 = alive0;
        B.lo = lo; B.hi = hi; B.res = res; B.fixed_const = fixed_const;
        B.simpson = simpson; B.dead = dead ? 1 : 0; B.vary = vary; B.np = UNI(r.npend); B.c0 = 0; B.nt = 0;
        B.chn = chn; B.s_in = s_in; B.s_out = s_out;
    }
    VLR_SYNC();
}
// tasks of the pending outer points [c0, c0 + kRows); returns false if the inner range is dead (no chains to run)
__device__ __forceinline__ bool bo_setup(Ctx& c, const Frame& f, RangeSt& r) {
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const BatchOuter& B = w->bo;
    const int lane = (((VLR_FRESH_MASK >> 6) & 1u) ? fresh_lane(c.lane) : (c.lane)), S = c.S;
    const int np = UNI(B.np), c0 = UNI(B.c0), s_in = UNI(B.s_in), s_out = UNI(B.s_out), chn = UNI(B.chn);
    const bool dead = UNI(B.dead) != 0;
    const int free_rows = kRows - c.nhold;  // held event-level chains keep the top rows (they ride along with this batch)
    const int nt = (np - c0) < free_rows ? (np - c0) : free_rows;
    PROF_ADD(c, 18);  // outer batch: entry (fixed samples) / delivery of the previous pass
    VLR_SYNC();
    if (lane == 0) w->bo.nt = nt;
    if (lane < nt) {
        const DevNode ch = ld_node(p.nodes + chn);
        const RangeV oorig{r.ostart, r.oend, r.olex, r.orex};
        const double x = r.pend[c0 + lane];
        ChainTask& T = w->task[lane];
        T.lo = B.lo; T.hi = B.hi; T.res = B.res;
        T.ostart = ch.vafs.start; T.oend = ch.vafs.end; T.olex = ch.vafs.lex; T.orex = ch.vafs.rex;
        T.simpson_n = B.simpson;
        T.contained = UNI(f.sv_contained) && range_contains(oorig, x);
        T.alive = alive_update(c, UNI_A(B.alive0), s_out, x) & NODE_ALIVE(ch);  // (& the groups that can contain a VAF of the inner node)
        int pidx = 0;
        for (int s = 0; s < S; ++s) {
            double v = (s == s_out) ? x : w->ops_vaf[s];
            c.tvaf[lane * S + s] = v;
            if (s != s_in) pidx += prior_class(p, s, v) * p.class_stride[s];
        }
        T.pidx = pidx;
        T.fixed = B.fixed_const;
        T.group = c.group; T.disc = c.disc & ~(1 << s_in);  // for the AFD log
        T.result = VLR_NEG_INF; T.haveBest = 0; T.n = 0; T.bestJ = VLR_NEG_INF; T.bestX = 0.0;
    }
    VLR_SYNC();
    PROF_ADD(c, 16);  // outer batch: task setup
    int vm = UNI(B.vary);
    while (vm && !dead) {
        int s = __builtin_ctz(vm);
        vm &= vm - 1;
        int by = p.by[s];
        if (lane < nt) {
            double a = c.tvaf[lane * S + s];
            double b = by >= 0 ? c.tvaf[lane * S + by] : 0.0;
            double al, be;
            alpha_beta(p, s, a, b, al, be);
            w->bpend[1][lane] = al;  // scratch: the row buffers are rewritten by the chain batch that follows
            w->bpend[2][lane] = be;
        }
        VLR_SYNC();
        int off = UNI(w->soff[s]), D = UNI(w->nkeep[s]);
        eval_pileup(c.coef + 2 * off, ecoef_of(c, s, off), D, (w->fastok >> s) & 1, nt, w->bpend[1], w->bpend[2], w->bpend[3], lane);
        VLR_SYNC();
        if (__builtin_expect(ones_any(c), 0) && ones_risk(c, s)) {
            if (lane < nt && is_all_ones(p, s, c.tvaf[lane * S + s], by >= 0 ? c.tvaf[lane * S + by] : 0.0)) {
                double Pm; int Em;
                ones_load(c, s, Pm, Em);
                w->bpend[3][lane] = ln_product_mantissa(Pm) + (double)Em * kLn2;
            }
            VLR_SYNC();
        }
#ifdef VLR_NO_RESCUE
        if (lane < nt) w->task[lane].fixed += w->bpend[3][lane];
#else
        if (lane < nt) w->task[lane].fixed += w->bpend[3][lane] + (double)kshift(c)[s] * kLn2;
#endif
        if (lane == 0) { w->work[0] += (unsigned long long)nt; w->work[1] += (unsigned long long)nt * (unsigned long long)D; }
        VLR_SYNC();
    }
    PROF_ADD(c, 17);  // outer batch: likelihoods of the samples that vary with the outer point
    c.bt_nt = nt;
    c.bt_inner = s_in;
    return !dead;
}
// hand the finished chains [c0, c0 + nt) to the outer frame; returns true while outer points are left
__device__ __forceinline__ bool bo_deliver(Ctx& c, const Frame& f, RangeSt& r, double* txo, double* tvo) {
    WaveSt* w = c.w;
    const BatchOuter& B = w->bo;
    const int lane = (((VLR_FRESH_MASK >> 7) & 1u) ? fresh_lane(c.lane) : (c.lane));
    const int np = UNI(B.np), c0 = UNI(B.c0), nt = UNI(B.nt), s_in = UNI(B.s_in), s_out = UNI(B.s_out);
    const bool dead = UNI(B.dead) != 0;
    PROF_ADD(c, 24);  // walk: resume up to the delivery
    VLR_SYNC();
    // lane i < nt fetches the results of chain i and records its outer point in one go; the row loop below then reads lanes
    // instead of making an LDS round trip per field and row
    const int li = lane < nt ? lane : 0;
    const ChainTask& Tl = w->task[li];
    const double xl = r.pend[c0 + li], resl = Tl.result, bJl = Tl.bestJ, bXl = Tl.bestX;
    const int hbl = Tl.haveBest, contl = Tl.contained, nl = Tl.n;
    const alive_t alivel = Tl.alive;
    const int tn0 = UNI(r.tn);
    if (lane < nt) { txo[tn0 + c0 + lane] = xl; tvo[tn0 + c0 + lane] = dead ? VLR_NEG_INF : resl; }
    PROF_ADD(c, 25);  // delivery: fetch + outer table
    for (int i = 0; i < nt; ++i) {
        const double x = lane_d(xl, i);
        VLR_SYNC();
        if (lane == 0) w->ops_vaf[s_out] = x;
        VLR_SYNC();
        if (__builtin_expect(dead, 0)) continue;
        const int hb = VLR_RDLANE(hbl, i), n_i = VLR_RDLANE(nl, i);
        if (__builtin_expect(c.replay != 0, 0)) {
            if (UNI(f.sv_mute) || table_has(txo, tn0 + c0 + i, x, lane)) continue;  // repeated outer VAF: same map keys
            afd_emit_row(c, i, s_in, n_i);
            continue;
        }
        if (hb & 1) map_consider(c, lane_d(bJl, i), s_in, lane_d(bXl, i));
        // rare: candidates for other groups / containment via another path (also of a visited excluded range end)
#ifdef VLR_WIDE_BUILD
        const alive_t al_i = ALIVE_OF(VLR_RDLANE((int)alivel, i), VLR_RDLANE((int)(alivel >> 32), i));
#else
        const alive_t al_i = VLR_RDLANE(alivel, i);
#endif
        const int co_i = VLR_RDLANE(contl, i);
        if (__builtin_expect(al_i != 0 || !co_i || (hb & 2), 0)) {
            const ChainTask& T = w->task[i];
            const RangeV io{uni_d(T.ostart), uni_d(T.oend), UNI(T.olex), UNI(T.orex)};
            scan_chain_candidates(c, c.rowX + i * c.cap, c.rowV + i * c.cap, n_i, io, co_i, al_i, s_in);
        }
    }
    const int c1 = c0 + nt;
    PROF_ADD(c, 26);  // delivery: MAP candidates per chain
    VLR_SYNC();
    if (c1 < np) {
        if (lane == 0) w->bo.c0 = c1;
        VLR_SYNC();
        return true;
    }
    PROF_ADD(c, 18);
    if (lane == 0) r.tn = r.tn + np;
    VLR_SYNC();
    return false;
}

// Event-level deferral: an event root whose path is a chain of single-valued Sample nodes ending in a leaf Range
// (tumor-normal: somatic_tumor, germline_het, germline_hom) contributes exactly one innermost chain.  Such chains of
// different events are collected and run together, one per DPP row (run_chain_batch); flush_deliver hands the
// integrals to the event accumulators and the MAP candidates to the event slots.
__device__ __forceinline__ void flush_deliver(Ctx& c, int rowmask, double* evM, double* evS, double bias_prior) {
    // (the chains were run by the event loop's single run_chain_batch site; a lone deferred chain takes the same path)
    WaveSt* w = c.w;
    rowmask = UNI(rowmask);
    VLR_SYNC();
    while (rowmask) {
        const int i = __builtin_ctz(rowmask);
        rowmask &= rowmask - 1;
        const ChainTask& T = w->task[i];
        const int u = UNI(T.u), s_in = UNI(T.inner);
        // restore the context of the deferred leaf: operands, event group, flags, and the slot's MAP candidate
        VLR_SYNC();
        if (c.lane < c.S) { w->ops_vaf[c.lane] = c.tvaf[i * c.S + c.lane]; w->curMapVaf[c.lane] = c.mapVaf[u * c.S + c.lane]; }
        VLR_SYNC();
        c.group = UNI(T.group); c.disc = UNI(T.disc); c.contained = UNI(T.contained); c.alive = UNI_A(T.alive); c.nlfc = 0;
        c.curJ = uni_d(c.mapJ[u]); c.curHyp = UNI(c.mapHyp[u]);
        const double dens = uni_d(T.result);
        TRC(c, 100, dens); TRC(c, 101, u); TRC(c, 102, i);
        if (dens != dens) c.status |= VLR_LOCUS_NAN;
        const int nq = UNI(T.n);
        if (__builtin_expect(c.replay != 0, 0)) afd_emit_row(c, i, s_in, nq);
        else {
            if (UNI(T.haveBest) & 1) map_consider(c, uni_d(T.bestJ), s_in, uni_d(T.bestX));
            if (__builtin_expect(c.alive != 0 || !c.contained || (UNI(T.haveBest) & 2), 0)) {
                const RangeV io{uni_d(T.ostart), uni_d(T.oend), UNI(T.olex), UNI(T.orex)};
                scan_chain_candidates(c, c.rowX + i * c.cap, c.rowV + i * c.cap, nq, io, c.contained, c.alive, s_in);
            }
        }
        double M = uni_d(evM[u]), Sx = uni_d(evS[u]);
        lse_add(M, Sx, bias_prior + dens);
        VLR_SYNC();
        if (c.lane == 0) { evM[u] = M; evS[u] = Sx; c.mapJ[u] = c.curJ; c.mapHyp[u] = c.curHyp; }
        if (c.lane < c.S) c.mapVaf[u * c.S + c.lane] = w->curMapVaf[c.lane];
        VLR_SYNC();
    }
}
// one chain task with its operands from row `from` (or the stash: from < 0) to row `to` (or the stash: to < 0)
__device__ __forceinline__ void move_task(Ctx& c, int from, int to) {
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 8) & 1u) ? fresh_lane(c.lane) : (c.lane));
    constexpr int NW = (int)(sizeof(ChainTask) / 8);
    static_assert(sizeof(ChainTask) % 8 == 0, "ChainTask is copied in 8-byte words");
    const double* src = (const double*)(from < 0 ? &w->stash : &w->task[from]);
    double* dst = (double*)(to < 0 ? &w->stash : &w->task[to]);
    const double* vs = from < 0 ? w->stash_vaf : c.tvaf + from * c.S;
    double* vd = to < 0 ? w->stash_vaf : c.tvaf + to * c.S;
    VLR_SYNC();
    const double a = src[lane < NW ? lane : 0], b = vs[lane < c.S ? lane : 0];
    VLR_SYNC();
    if (lane < NW) dst[lane] = a;
    if (lane < c.S) vd[lane] = b;
    VLR_SYNC();
}

// Compiled chain root (DevFastRoot, vlr_plan.h): what walk_root does in the probe pass for a root that is a chain of single-valued
// Sample nodes ending in a leaf Range node — clear-ref shortcuts of every node on the way (modes/generic.rs:270-347), operands,
// the leaf's observable range, fixed-sample likelihoods, the deferred ChainTask — from the plan's record instead of the tree.
// Returns 0: the density is ln 0 (a node is dead: nothing to add to the event), 1: deferred as task c.ndef - 1,
// 2: leave the root to the general pass (another integrated sample than the chains collected so far / tables above 64 entries).
__device__ __forceinline__ int fast_chain_root(Ctx& c, const DevFastRoot* fr) {
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    const int lane = (((VLR_FRESH_MASK >> 9) & 1u) ? fresh_lane(c.lane) : (c.lane)), S = c.S;
    const int n_fixed = ldc(&fr->n_fixed), inner = ldc(&fr->inner);
    // the fixed values, one per lane (lane k < n_fixed: node k of the path)
    const int kf = lane < n_fixed ? lane : 0;
    const int fs = fr->fsample[kf];
    const double fv = fr->fvaf[kf];
    // clear_ref && every VAF of the node above zero: the node (and the root) is dead — Set 294-330, singleton Range 342-347
    const bool dead_l = lane < n_fixed && w->nkeep[fs] > 10 && w->all_posref[fs] != 0 && fv > 0.0;
    const int n_obs = UNI(w->nkeep[inner]);
    const bool clear_in = n_obs > 10 && UNI((int)w->all_posref[inner]) != 0;
    const RangeV vr{ldc(&fr->start), ldc(&fr->end), ldc(&fr->lex), ldc(&fr->rex)};
    if (__ballot(dead_l) != 0ull || (clear_in && vr.start > 0.0)) return 0;
    if (c.cap > 64 || (c.ndef > 0 && UNI(w->task[0].inner) != inner)) return 2;
    VLR_SYNC();
    if (lane < n_fixed) w->ops_vaf[fs] = fv;
    VLR_SYNC();
    const double res = p.resolution[inner];
    const double lo = uni_d(observable_min(vr, n_obs)), hi = uni_d(observable_max(vr, n_obs));
    const int simpson = ((hi - lo) < res) ? 3 : (n_obs < 5 ? 11 : 0);  // 367-394
    double fixed = 0.0;
    for (int s2 = 0; s2 < S; ++s2) {
        const int by = p.by[s2];
        if (!(s2 == inner || by == inner)) fixed += sample_lik(c, s2, w->ops_vaf[s2], by >= 0 ? w->ops_vaf[by] : 0.0);
    }
    const int row = c.ndef;
    VLR_SYNC();
    if (lane == 0) {
        ChainTask& T = w->task[row];
        T.lo = lo; T.hi = hi; T.res = res; T.ostart = vr.start; T.oend = vr.end; T.olex = vr.lex; T.orex = vr.rex;
        T.simpson_n = simpson; T.fixed = fixed; T.pidx = ldc(&fr->pidx); T.contained = 1; T.alive = ALIVE_OF(ldc(&fr->alive), ldc(&fr->alive_hi));
        T.result = VLR_NEG_INF; T.haveBest = 0; T.n = 0; T.bestJ = VLR_NEG_INF; T.bestX = 0.0;
        T.group = c.group; T.disc = ldc(&fr->disc); T.inner = inner; T.u = c.defer_slot;
    }
    if (lane < S) c.tvaf[row * S + lane] = w->ops_vaf[lane];
    VLR_SYNC();
    c.ndef = row + 1;
    c.afd_mute = 0;
    return 1;
}

// node id of the single child of `fnode` if that child is a leaf Sample node with a proper Range spectrum, else -1
__device__ __forceinline__ int leaf_range_child(const DevPlan& p, int fnode) {
    const DevNode* f = p.nodes + fnode;
    if (ldc(&f->n_children) != 1) return -1;
    const int chn = ldc(p.child_index + ldc(&f->child_off));
    const DevNode* ch = p.nodes + chn;
    const bool ok = ldc(&ch->kind) == VLR_NODE_SAMPLE && ldc(&ch->vafs.kind) == 1 && ldc(&ch->n_children) == 0 &&
                    ldc(&ch->vafs.start) != ldc(&ch->vafs.end);
    return ok ? chn : -1;
}

// GenericPosterior::density (modes/generic.rs:190-423) for one (hypothesis, root): explicit-stack walk.
__device__ __forceinline__ double walk_root(Ctx& c, int root, int resume) {
    const DevPlan& p = *c.plan;
    WaveSt* w = c.w;
    int sp = 0, node = UNI(root), nrange = 0;
    enum { PC_DESCEND, PC_SUB, PC_RETURN, PC_RANGE_ISSUE, PC_BO_PRE, PC_BO_POST } pc = PC_DESCEND;
    double rv = VLR_NEG_INF;
    bool skip_record = false;
    if (resume) {  // the event loop ran the chain batch this walk asked for
        const WalkSave& k = w->wk;
        sp = UNI(k.sp); node = UNI(k.node); nrange = UNI(k.nrange); skip_record = UNI(k.skip_record) != 0; rv = uni_d(k.rv);
        pc = PC_BO_POST;
    } else {
        c.present = 0; c.disc = 0; c.nlfc = 0; c.contained = 1; c.afd_mute = 0;
        c.alive = ALIVE_FULL(p.n_named + 1) & ~ALIVE_BIT(c.group);  // n_named <= 30 (wide build: 62): no shift reaches the sign bit
    }
    for (;;) {
        if (pc == PC_DESCEND) {
            const DevNode nd = ld_node(p.nodes + node);
            if (nd.kind == VLR_NODE_LFC) {  // 233-244
                if (c.nlfc < kMaxLfc) {
                    VLR_SYNC();
                    if (c.lane == 0) {
                        w->lfc_a[c.nlfc] = nd.sample; w->lfc_b[c.nlfc] = nd.sample_b;
                        w->lfc_cmp[c.nlfc] = nd.cmp; w->lfc_val[c.nlfc] = nd.lfc_value;
                    }
                    VLR_SYNC();
                    c.nlfc++;
                }
                pc = PC_SUB;
            } else if (nd.kind == VLR_NODE_FALSE) { rv = VLR_NEG_INF; pc = PC_RETURN; }
            else if (nd.kind == VLR_NODE_TRUE) { rv = 0.0; pc = PC_RETURN; }
            else if (nd.kind == VLR_NODE_VARIANT) {  // 398-420
                bool go;
                if (c.has_snv) {
                    bool contains = iupac_contains(nd.refbase, c.refbase) && iupac_contains(nd.altbase, c.altbase);
                    go = !((nd.positive && !contains) || (!nd.positive && contains));
                } else go = !nd.positive;
                if (go) pc = PC_SUB; else { rv = VLR_NEG_INF; pc = PC_RETURN; }
            } else {  // Sample (247-397)
                int s = nd.sample;
                RangeV bounds;
                bool have_bounds = ops_lfc_bounds(c, s, bounds);
                int n_obs = w->nkeep[s];
                bool clear_ref = n_obs > 10 && w->all_posref[s];  // 270-291
                bool is_set = nd.vafs.kind == 0;
                RangeV vr{nd.vafs.start, nd.vafs.end, nd.vafs.lex, nd.vafs.rex};
                bool dead = have_bounds && range_is_empty(bounds);  // 262-268
                int ncand = 0;
                bool as_set = false;
                if (!dead) {
                    if (is_set) {  // 294-330
                        bool all_pos = true;
                        for (int i = 0; i < nd.vafs.set_len; ++i) all_pos = all_pos && (ldc(p.vafs + nd.vafs.set_off + i) > 0.0);
                        if (clear_ref && all_pos) dead = true;
                        else {
                            VLR_SYNC();
                            for (int i = 0; i < nd.vafs.set_len && ncand < p.max_set; ++i) {
                                double v = ldc(p.vafs + nd.vafs.set_off + i);
                                if (!have_bounds || range_contains(bounds, v)) {
                                    if (c.lane == 0) c.setv[s * p.max_set + ncand] = v;
                                    ncand++;
                                }
                            }
                            VLR_SYNC();
                            as_set = true;
                            if (ncand == 0) dead = true;  // ln_sum_exp of nothing
                        }
                    } else {  // 331-395
                        if (have_bounds) vr = range_intersect(vr, bounds);
                        if (range_is_empty(vr)) dead = true;
                        else if (clear_ref && vr.start > 0.0) dead = true;
                        else if (range_is_singleton(vr)) {
                            VLR_SYNC();
                            if (c.lane == 0) c.setv[s * p.max_set] = vr.start;
                            VLR_SYNC();
                            ncand = 1;
                            as_set = true;
                        }
                    }
                }
                PROF_ADD(c, 32);  // walk: Sample node, bounds / candidates
                if (dead) { rv = VLR_NEG_INF; pc = PC_RETURN; }
                else if (sp >= c.nframes) { c.status |= VLR_LOCUS_TABLE_FULL; rv = __builtin_nan(""); pc = PC_RETURN; }
                else {
                    VLR_SYNC();
                    Frame& f = c.frames[sp];
                    if (c.lane == 0) {
                        f.node = node; f.iter = 0; f.accM = VLR_NEG_INF; f.accS = 0.0;
                        f.sv_present = c.present; f.sv_disc = c.disc; f.sv_nlfc = c.nlfc; f.sv_contained = c.contained;
                        // every operand set below this frame takes sample s from this node: groups whose spectra for s miss the
                        // node's spectrum altogether (static, DevNode::alive_mask) cannot contain any of them
                        f.sv_alive = c.alive & NODE_ALIVE(nd); f.sv_mute = c.afd_mute;
                    }
                    if (c.defer_ok && (as_set ? (ncand > 1) : (nd.n_children != 0))) {
                        c.deferred = 2;  // probe pass: not a single-chain root, evaluate in the second pass
                        return 0.0;
                    }
                    if (as_set) {
                        if (c.lane == 0) { f.kind = FK_SET; f.n = ncand; w->ops_vaf[s] = c.setv[s * p.max_set]; }
                        VLR_SYNC();
                        sp++;
                        c.present |= (1 << s);
                        c.disc |= (1 << s);
                        c.contained = UNI(f.sv_contained) && spectrum_contains(nd.vafs, p.vafs, w->ops_vaf[s]);
                        c.alive = alive_update(c, UNI_A(f.sv_alive) & NODE_ALIVE(nd), s, w->ops_vaf[s]);
                        pc = PC_SUB;
                    } else if (nrange >= p.max_range_depth || nrange >= kMaxRangeDepth) {
                        c.status |= VLR_LOCUS_TABLE_FULL; rv = __builtin_nan(""); pc = PC_RETURN;
                    } else {
                        RangeSt& r = c.rs[nrange];
                        double res = p.resolution[s];
                        double min_vaf = observable_min(vr, n_obs);
                        double max_vaf = observable_max(vr, n_obs);
                        // the leaf Range child of this frame, if it has one (looked up once: every round of the frame asks)
                        const int lchild = (nd.n_children != 0) ? leaf_range_child(p, node) : -1;
                        if (c.lane == 0) {
                            f.kind = FK_RANGE; f.slot = nrange; f.n = lchild;
                            r.lo = min_vaf; r.hi = max_vaf; r.res = res;
                            r.ostart = nd.vafs.start; r.oend = nd.vafs.end; r.olex = nd.vafs.lex; r.orex = nd.vafs.rex;
                            r.have_first = 0; r.have_mid = 0; r.tn = 0; r.sample = s; r.leaf = (nd.n_children == 0);
                            int simpson = ((max_vaf - min_vaf) < res) ? 3 : (n_obs < 5 ? 11 : 0);  // 367-394
                            r.simpson_n = simpson;
                            if (simpson) {
                                // density is evaluated for interior points first, then a, b (bio); table order = grid order
                                double step = (max_vaf - min_vaf) / (double)(simpson - 1);
                                for (int i = 0; i < simpson; ++i) r.pend[i] = lin_pt(min_vaf, step, (double)i);
                                r.pend[0] = min_vaf; r.pend[simpson - 1] = max_vaf;
                                r.npend = simpson; r.phase = RP_SIMPSON;
                            } else {
                                r.pend[0] = min_vaf; r.pend[1] = max_vaf; r.npend = 2; r.phase = RP_INIT;
                            }
                        }
                        VLR_SYNC();
                        sp++;
                        nrange++;
                        c.present |= (1 << s);
                        c.disc &= ~(1 << s);
                        PROF_ADD(c, 33);  // walk: Range frame set-up (observable bounds)
                        pc = PC_RANGE_ISSUE;
                    }
                }
            }
        } else if (pc == PC_SUB) {  // subdensity (199-230)
            const DevNode nd = ld_node(p.nodes + node);
            if (nd.n_children == 0) { rv = leaf_joint(c); pc = PC_RETURN; }
            else if (nd.n_children == 1) { node = ldc(p.child_index + nd.child_off); pc = PC_DESCEND; }
            else if (sp >= c.nframes) { c.status |= VLR_LOCUS_TABLE_FULL; rv = __builtin_nan(""); pc = PC_RETURN; }
            else if (c.defer_ok) { c.deferred = 2; return 0.0; }  // probe pass: branching root is not a single chain
            else {
                VLR_SYNC();
                Frame& f = c.frames[sp];
                if (c.lane == 0) {
                    f.kind = FK_BRANCH; f.node = node; f.iter = 0; f.n = nd.n_children; f.accM = VLR_NEG_INF; f.accS = 0.0;
                    f.sv_present = c.present; f.sv_disc = c.disc; f.sv_nlfc = c.nlfc; f.sv_contained = c.contained;
                    f.sv_alive = c.alive; f.sv_mute = c.afd_mute;
                }
                VLR_SYNC();
                sp++;
                node = ldc(p.child_index + nd.child_off);
                pc = PC_DESCEND;
            }
        } else if (pc == PC_RANGE_ISSUE) {
            Frame& f = c.frames[sp - 1];
            const int fslot = UNI(f.slot), fnode = UNI(f.node);
            RangeSt& r = c.rs[fslot];
            double* tx = c.tabX + fslot * c.cap;
            if (UNI(r.tn) + (UNI(r.npend) - UNI(f.iter)) > c.cap) {  // room for the points of this round still to be recorded
                c.status |= VLR_LOCUS_TABLE_FULL;
                rv = __builtin_nan("");
                c.present = UNI(f.sv_present); c.disc = UNI(f.sv_disc); c.nlfc = UNI(f.sv_nlfc); c.contained = UNI(f.sv_contained); c.alive = UNI_A(f.sv_alive); c.afd_mute = UNI(f.sv_mute);
                sp--; nrange--;
                pc = PC_RETURN;
            } else if (UNI(r.leaf)) {
                // innermost chain: evaluate all pending points at once, loop the state machine here
                c.present = UNI(f.sv_present) | (1 << UNI(r.sample));
                c.nlfc = UNI(f.sv_nlfc);
                c.contained = UNI(f.sv_contained);
                c.alive = UNI_A(f.sv_alive);
                if (c.defer_ok && (c.cap > 64 || c.nlfc != 0 || (c.ndef > 0 && UNI(w->task[0].inner) != UNI(r.sample)))) {
                    c.deferred = 2;
                    return 0.0;
                }
                PROF_ADD(c, 35);  // walk: up to the leaf Range
                if (c.defer_ok) {
                    // defer: this root is exactly one innermost chain (all enclosing frames are single-valued)
                    const int inner = UNI(r.sample), S = c.S, row = c.ndef;
                    double fixed = 0.0;
                    int pidx = 0;
                    for (int s2 = 0; s2 < S; ++s2) {
                        int by = p.by[s2];
                        if (!(s2 == inner || by == inner)) fixed += sample_lik(c, s2, w->ops_vaf[s2], by >= 0 ? w->ops_vaf[by] : 0.0);
                        if (s2 != inner) pidx += prior_class(p, s2, w->ops_vaf[s2]) * p.class_stride[s2];
                    }
                    VLR_SYNC();
                    if (c.lane == 0) {
                        ChainTask& T = w->task[row];
                        T.lo = r.lo; T.hi = r.hi; T.res = r.res; T.ostart = r.ostart; T.oend = r.oend; T.olex = r.olex; T.orex = r.orex;
                        T.simpson_n = r.simpson_n; T.fixed = fixed; T.pidx = pidx; T.contained = c.contained; T.alive = c.alive;
                        T.result = VLR_NEG_INF; T.haveBest = 0; T.n = 0; T.bestJ = VLR_NEG_INF; T.bestX = 0.0;
                        T.group = c.group; T.disc = c.disc; T.inner = inner; T.u = c.defer_slot;
                    }
                    if (c.lane < S) c.tvaf[row * S + c.lane] = w->ops_vaf[c.lane];
                    VLR_SYNC();
                    c.ndef = row + 1;
                    c.deferred = 1;
                    PROF_ADD(c, 34);  // walk: deferred leaf chain (fixed likelihoods, prior index, task)
                    return 0.0;
                }
                rv = run_leaf_chain(c, r, c.rowX, c.rowV);
                c.present = UNI(f.sv_present); c.disc = UNI(f.sv_disc); c.nlfc = UNI(f.sv_nlfc); c.contained = UNI(f.sv_contained); c.alive = UNI_A(f.sv_alive); c.afd_mute = UNI(f.sv_mute);
                sp--; nrange--;
                pc = PC_RETURN;
            } else if (c.cap <= 64 && UNI(f.iter) == 0 && UNI(f.sv_nlfc) == 0 && UNI(f.n) >= 0) {
                // outer chain over a leaf Range child: all pending points at once, kRows inner chains per pass
                c.present = UNI(f.sv_present) | (1 << UNI(r.sample));
                c.disc = UNI(f.sv_disc) & ~(1 << UNI(r.sample));
                c.nlfc = UNI(f.sv_nlfc);
                // the frame's constants (inner range, fixed likelihoods, ...) are set up in its first round only; later rounds
                // just rewind the point counters (no other outer frame can run between the rounds of this one: its children
                // are leaf chains)
                PROF_ADD(c, 29);  // outer: round issue
                if (UNI(r.tn) == 0) { bo_begin(c, r, UNI(f.n), UNI_A(f.sv_alive)); PROF_ADD(c, 30); }
                else {
                    VLR_SYNC();
                    if (c.lane == 0) { BatchOuter& B = w->bo; B.np = r.npend; B.c0 = 0; B.nt = 0; }
                    VLR_SYNC();
                }
                pc = PC_BO_PRE;
            } else {
                // outer chain: one point at a time through the subtree
                int it = UNI(f.iter);
                double x = uni_d(r.pend[it]);
                c.present = UNI(f.sv_present) | (1 << UNI(r.sample));
                c.disc = UNI(f.sv_disc) & ~(1 << UNI(r.sample));
                c.nlfc = UNI(f.sv_nlfc);
                RangeV orig{r.ostart, r.oend, r.olex, r.orex};
                c.contained = UNI(f.sv_contained) && range_contains(orig, x);
                c.alive = alive_update(c, UNI_A(f.sv_alive), UNI(r.sample), x);
                if (c.replay) c.afd_mute = UNI(f.sv_mute) || table_has(tx, UNI(r.tn), x, c.lane);
                VLR_SYNC();
                if (c.lane == 0) w->ops_vaf[UNI(r.sample)] = x;
                VLR_SYNC();
                node = fnode;
                pc = PC_SUB;
            }
        } else if (pc == PC_BO_PRE) {
            Frame& f = c.frames[sp - 1];
            RangeSt& r = c.rs[UNI(f.slot)];
            if (bo_setup(c, f, r)) {
                VLR_SYNC();
                if (c.lane == 0) { WalkSave& k = w->wk; k.sp = sp; k.node = node; k.nrange = nrange; k.skip_record = skip_record ? 1 : 0; k.rv = rv; }
                VLR_SYNC();
                c.need_batch = 1;
                return 0.0;
            }
            pc = PC_BO_POST;
        } else if (pc == PC_BO_POST) {
            Frame& f = c.frames[sp - 1];
            const int fslot = UNI(f.slot);
            RangeSt& r = c.rs[fslot];
            double* tx = c.tabX + fslot * c.cap;
            double* tv = c.tabV + fslot * c.cap;
            if (bo_deliver(c, f, r, tx, tv)) pc = PC_BO_PRE;
            else {
                // all points of the round are recorded (f.iter stays 0 in batched rounds): advance the outer chain right here
                // (what PC_RETURN does for a frame whose points come back one by one)
                const bool done = range_advance(c, r, tx, tv);
                VLR_SYNC();
                PROF_ADD(c, 27);  // outer: range_advance
                if (!done) pc = PC_RANGE_ISSUE;
                else {
                    rv = range_finish(c, r, tx, tv);
                    PROF_ADD(c, 28);  // outer: range_finish
                    c.present = UNI(f.sv_present); c.disc = UNI(f.sv_disc); c.nlfc = UNI(f.sv_nlfc); c.contained = UNI(f.sv_contained); c.alive = UNI_A(f.sv_alive); c.afd_mute = UNI(f.sv_mute);
                    sp--; nrange--;
                    pc = PC_RETURN;
                }
            }
        } else {  // PC_RETURN: hand rv to the enclosing frame
            rv = uni_d(rv);
            if (sp == 0) return rv;
            Frame& f = c.frames[sp - 1];
            if (UNI(f.kind) == FK_RANGE) {
                const int fslot = UNI(f.slot);
                RangeSt& r = c.rs[fslot];
                double* tx = c.tabX + fslot * c.cap;
                double* tv = c.tabV + fslot * c.cap;
                VLR_SYNC();
                if (c.lane == 0 && !skip_record) { tx[r.tn] = r.pend[f.iter]; tv[r.tn] = rv; r.tn = r.tn + 1; f.iter = f.iter + 1; }
                skip_record = false;
                VLR_SYNC();
                if (UNI(f.iter) < UNI(r.npend)) { pc = PC_RANGE_ISSUE; }
                else {
                    bool done = range_advance(c, r, tx, tv);
                    VLR_SYNC();
                    if (c.lane == 0) f.iter = 0;
                    VLR_SYNC();
                    if (!done) { pc = PC_RANGE_ISSUE; }
                    else {
                        rv = range_finish(c, r, tx, tv);
                        c.present = UNI(f.sv_present); c.disc = UNI(f.sv_disc); c.nlfc = UNI(f.sv_nlfc); c.contained = UNI(f.sv_contained); c.alive = UNI_A(f.sv_alive); c.afd_mute = UNI(f.sv_mute);
                        sp--; nrange--;
                        pc = PC_RETURN;
                    }
                }
            } else {  // SET or BRANCH: ln_sum_exp over members / children (203-217, 319-328)
                double M = f.accM, S = f.accS;
                lse_add(M, S, rv);
                int it = UNI(f.iter) + 1;
                VLR_SYNC();
                if (c.lane == 0) { f.accM = M; f.accS = S; f.iter = it; }
                VLR_SYNC();
                c.present = UNI(f.sv_present); c.disc = UNI(f.sv_disc); c.nlfc = UNI(f.sv_nlfc); c.contained = UNI(f.sv_contained); c.alive = UNI_A(f.sv_alive); c.afd_mute = UNI(f.sv_mute);
                if (it < UNI(f.n)) {
                    const int fnode = UNI(f.node);
                    const DevNode nd = ld_node(p.nodes + fnode);
                    if (UNI(f.kind) == FK_SET) {
                        int s = nd.sample;
                        VLR_SYNC();
                        if (c.lane == 0) w->ops_vaf[s] = c.setv[s * p.max_set + it];
                        VLR_SYNC();
                        c.present |= (1 << s);
                        c.disc |= (1 << s);
                        c.contained = UNI(f.sv_contained) && spectrum_contains(nd.vafs, p.vafs, w->ops_vaf[s]);
                        c.alive = alive_update(c, UNI_A(f.sv_alive), s, w->ops_vaf[s]);
                        node = fnode;
                        pc = PC_SUB;
                    } else {
                        node = ldc(p.child_index + nd.child_off + it);
                        pc = PC_DESCEND;
                    }
                } else {
                    rv = (M != M) ? M : lse_value(M, S);
                    sp--;
                    pc = PC_RETURN;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FORMAT/AFD from the log of the call pass (calling.rs:889-928): one wave per locus filters the logged operand sets with the
// rules of afd_consider — equal to the MAP on all other samples (VAF and is_discrete), contained in the best event's tree
// with the sample excluded, one entry per operand key (afd_finish).
//   1. all record headers at once (directory): lane r classifies record r — how many samples other than the integrated one
//      differ from the MAP; a record whose key (sample, is_discrete flags, operands, l2fc terms) repeats an earlier one is a
//      re-visit of the same map keys (an outer chain evaluating a VAF twice) and is dropped;
//   2. records that agree with the MAP everywhere else (the chain the MAP sits on, typically one): one `contains` walk for the
//      record, then its whole table goes to the list of its sample with coalesced stores; the entry at the MAP VAF itself and
//      the matching entries of the records that differ in exactly one sample take the scalar afd_consider path;
//   3. discrete leaves: classified one per lane, matches through afd_consider.
__global__ void __launch_bounds__(64) vlr_afd_kernel(const DevPlan plan_arg, DevBatch batch, DevResults out) {
    __shared__ WaveSt wst;
    extern __shared__ double sh_seen[];  // [S][seen_cap] (VAF, l2fc key) pairs (dynamic: the plan's largest Set spectrum sets the size)
    __shared__ double sh_mapv[kLdsSamples];
    __shared__ int sh_nseen[kLdsSamples];
    __shared__ int sh_cnt[kLdsSamples];
    const DevPlan& p = plan_arg;
    const int lane = threadIdx.x;
    const int64_t locus = blockIdx.x;
    if (locus >= batch.n_loci) return;
    const int S = p.S;
    const double* lg = out.afd_log + (size_t)locus * (size_t)out.afd_log_stride;
    if (__double_as_longlong(lg[0]) < 0) return;  // overflowed: the replay launch handles this locus
    bool have = true;
    for (int s = 0; s < S; ++s) { double v = out.map_vaf[locus * S + s]; if (v != v) have = false; }
    bool art = false;
    if (out.map_bias) for (int i = 0; i < VLR_N_BIAS; ++i) art = art || out.map_bias[locus * VLR_N_BIAS + i] != 0;
    if (!have || art) return;  // AFD only exists for a non-artifact MAP
    Ctx c;
    c.plan = &plan_arg; c.w = &wst; c.lane = lane; c.S = S;
    c.afd_seen = sh_seen; c.mapv = sh_mapv; c.afd_nseen = sh_nseen;
    c.seen_cap = p.max_set > 16 ? p.max_set : 16;
    c.replay = 1; c.hyp = 0; c.afd_mute = 0; c.locus = locus; c.outp = &out; c.status = 0; c.lg = nullptr; c.lg_pos = -1; c.lg_cap = 0; c.lg_nrec = 0;
    c.vt = 0; c.has_snv = 0; c.refbase = 0; c.altbase = 0;
    if (lane < S) { sh_mapv[lane] = out.map_vaf[locus * S + lane]; sh_nseen[lane] = 0; sh_cnt[lane] = 0; }
    c.afd_cnt = sh_cnt;
    VLR_SYNC();
    const int be = UNI(out.best_event[locus]);
    c.mapGroup = (be == 0) ? 0 : ((be - 1) / 2 + 1);
    c.mapDisc = UNI((int)out.map_disc[locus]);
    c.marginal = uni_d(out.ln_marginal[locus]);
    const int nrec = (int)__double_as_longlong(lg[1]);
    // ---- 1. classify the records, one per lane (headers, operands and three probes of the table go through LDS so that the
    // comparison with the earlier records does not chase global memory)
    __shared__ long long sh_hdr[kLogDir];
    __shared__ double sh_ops[kLogDir][kLdsSamples];
    __shared__ double sh_probe[kLogDir][3];
    __shared__ double sh_x[kTableCap];
    int at = 0, n = 0, s_in = -1, disc = 0, nl = 0, kind = 0, grp = 0, mism = 99;
    long long h = 0;
    if (lane < nrec) {
        at = (int)__double_as_longlong(lg[2 + lane]);
        h = __double_as_longlong(lg[at]);
        n = (int)(h & 0xffff); s_in = (int)((h >> 16) & 0x1f) - 1; disc = (int)((h >> 21) & 0xffff); grp = (int)((h >> 37) & 0xff);
        nl = (int)((h >> 45) & 0xf); kind = (int)((h >> 49) & 3);
        sh_hdr[lane] = h;
        if (kind != 3) {
            for (int s = 0; s < S; ++s) sh_ops[lane][s] = lg[at + 1 + s];
            const double* X = lg + at + 1 + S + 2 * nl;
            // two chains with equal keys are the same chain iff they integrate the same interval: first, second and last point
            sh_probe[lane][0] = X[0]; sh_probe[lane][1] = (kind == 1 && n > 1) ? X[1] : 0.0; sh_probe[lane][2] = (kind == 1) ? X[n - 1] : 0.0;
        }
    }
    VLR_SYNC();
    if (lane < nrec && kind != 3) {
        mism = 0;
        for (int s = 0; s < S; ++s) {
            if (s == s_in) continue;
            if (!(sh_ops[lane][s] == sh_mapv[s] && (((disc >> s) & 1) == ((c.mapDisc >> s) & 1)))) mism++;
        }
        // the same map keys as an earlier record? (an outer chain that evaluates a VAF twice runs the inner chain twice)
        for (int r = 0; r < lane && mism <= 1; ++r) {
            const long long h2 = sh_hdr[r];
            if (((h ^ h2) & ~(0xffll << 37)) != 0) continue;  // sample, flags, l2fc count, kind, table length (event group aside)
            bool same = true;
            for (int s = 0; s < S; ++s) same = same && (s == s_in || sh_ops[lane][s] == sh_ops[r][s]);
            if (kind == 1) same = same && sh_probe[lane][0] == sh_probe[r][0] && sh_probe[lane][1] == sh_probe[r][1] && sh_probe[lane][2] == sh_probe[r][2];
            if (same && nl > 0) {
                const int at2 = (int)__double_as_longlong(lg[2 + r]);
                for (int k = 0; k < 2 * nl; ++k) same = same && (__double_as_longlong(lg[at + 1 + S + k]) == __double_as_longlong(lg[at2 + 1 + S + k]));
            }
            if (same) mism = 98;
        }
    }
    // ---- 1b. chain records that differ from the MAP in exactly one other sample (the inner chains of all outer points but the
    // MAP's: two dozen per tumor-normal locus) contribute ONE entry, at the table's first x equal to the MAP VAF of the integrated
    // sample.  Found here for all of them with four coalesced table loads in flight and one gather of the values, instead of three
    // dependent memory round trips per record in the loop below.  (Tables above 64 entries or with l2fc terms take that loop.)
    int hitq = -2;          // lane r: -2 = record r is not handled here, -1 = no such x, else its index
    double hitv = 0.0;
    for (int r0 = 0; r0 < nrec; r0 += 4) {
        double xs[4];
        bool fastr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u;
            fastr[u] = false; xs[u] = __builtin_nan("");
            if (r < nrec) {
                const int k_r = VLR_RDLANE(kind, r), m_r = VLR_RDLANE(mism, r);
                const int n_r = VLR_RDLANE(n, r), nl_r = VLR_RDLANE(nl, r);
                const int at_r = VLR_RDLANE(at, r);  // (read where every lane is on: lane r itself is off below when r >= n_r — round 6, EXEC assert build)
                fastr[u] = k_r == 1 && m_r == 1 && n_r <= 64 && nl_r == 0;
                if (fastr[u] && lane < n_r) xs[u] = lg[at_r + 1 + S + lane];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u;
            if (r < nrec && fastr[u]) {
                const double mxr = uni_d(sh_mapv[VLR_RDLANE(s_in, r)]);
                const unsigned long long hit = __ballot(xs[u] == mxr);
                if (lane == r) hitq = hit ? (int)__builtin_ctzll(hit) : -1;
            }
        }
    }
    if (hitq >= 0) hitv = lg[at + 1 + S + n + hitq];
    // ---- 2./3. records in log order
    for (int r = 0; r < nrec; ++r) {
        const int k_r = VLR_RDLANE(kind, r), m_r = VLR_RDLANE(mism, r);
        if (k_r != 3 && m_r > 1) continue;
        const int at_r = VLR_RDLANE(at, r), n_r = VLR_RDLANE(n, r), sin_r = VLR_RDLANE(s_in, r);
        const int nl_r = VLR_RDLANE(nl, r);
        c.group = VLR_RDLANE(grp, r);
        c.disc = VLR_RDLANE(disc, r); c.nlfc = nl_r;
        const int pay = at_r + 1 + S + 2 * nl_r;
        {
            const int hq = VLR_RDLANE(hitq, r);
            if (hq != -2) {  // handled by 1b: operands from LDS, value from the gather
                if (hq < 0) continue;
                VLR_SYNC();
                if (lane < S) wst.ops_vaf[lane] = sh_ops[r][lane];
                VLR_SYNC();
                const double hv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(hitv), r), __builtin_amdgcn_readlane(__double2loint(hitv), r));
                afd_consider(c, hv, sin_r, uni_d(sh_mapv[sin_r]), -1);
                continue;
            }
        }
        if (k_r == 3) {  // discrete leaves of one root: n_r x (S VAFs, joint)
            const double* L = lg + at_r + 1;
            for (int q0 = 0; q0 < n_r; q0 += 64) {
                const int q = q0 + lane;
                int mm = 99;
                if (q < n_r) {
                    mm = 0;
                    for (int s = 0; s < S; ++s)
                        if (!(L[q * (S + 1) + s] == sh_mapv[s] && ((c.mapDisc >> s) & 1))) mm++;
                }
                unsigned long long todo = __ballot(mm <= 1);
                while (todo) {
                    const int qq = q0 + __builtin_ctzll(todo);
                    todo &= todo - 1;
                    VLR_SYNC();
                    if (lane < S) wst.ops_vaf[lane] = L[qq * (S + 1) + lane];
                    VLR_SYNC();
                    afd_consider(c, uni_d(L[qq * (S + 1) + S]), -1, 0.0);
                }
            }
            continue;
        }
        VLR_SYNC();
        if (lane < S) wst.ops_vaf[lane] = lg[at_r + 1 + lane];
        if (lane < nl_r) {
            const long long t = __double_as_longlong(lg[at_r + 1 + S + 2 * lane]);
            wst.lfc_a[lane] = (int)(t & 0xff); wst.lfc_b[lane] = (int)((t >> 8) & 0xff); wst.lfc_cmp[lane] = (int)((t >> 16) & 0xff);
            wst.lfc_val[lane] = lg[at_r + 1 + S + 2 * lane + 1];
        }
        VLR_SYNC();
        if (k_r == 2) { afd_consider(c, uni_d(lg[pay]), -1, 0.0); continue; }
        const double* X = lg + pay;
        const double* V = X + n_r;
        const double mx = uni_d(sh_mapv[sin_r]);
        const bool bulk = m_r == 0 && group_contains(c, c.mapGroup, sin_r, 0.0, sin_r);  // the integrated sample is excluded: x does not matter
        const long long lkey_r = UNI64(lfc_ctx_key(c));
        // the x values of the record go to LDS in one coalesced round (table capacity <= kTableCap = 128): the first-occurrence test
        // below reads every earlier x once per entry — as uniform global loads that was one memory round trip per table entry
        VLR_SYNC();
        for (int i = lane; i < n_r && i < kTableCap; i += 64) sh_x[i] = X[i];
        VLR_SYNC();
        for (int q0 = 0; q0 < n_r; q0 += 64) {
            const int q = q0 + lane;
            const bool on = q < n_r;
            const double xq = on ? (q < kTableCap ? sh_x[q] : X[q]) : __builtin_nan(""), vq = on ? V[q] : 0.0;
            // one map key per revisited point of the chain: first occurrence only
            bool dup = false;
            for (int j = 0; j < q0 + 64 && j < n_r; ++j) dup = dup || (j < q && (j < kTableCap ? sh_x[j] : X[j]) == xq);
            if (bulk) {
                const bool emit = on && !dup;
                const unsigned long long em = __ballot(emit);
                const int64_t slot = locus * S + sin_r;
                int base = UNI(sh_cnt[sin_r]);
                VLR_WAVE_FENCE();
                if (lane == 0 && em) sh_cnt[sin_r] = base + __popcll(em);
                VLR_WAVE_FENCE();
                const int idx = base + __popcll(em & ((1ull << lane) - 1ull));
                if (emit && idx < out.afd_capacity) {
                    out.afd_vaf[slot * out.afd_capacity + idx] = xq;  // continuous operand: no discrete marker
                    out.afd_lnprob[slot * out.afd_capacity + idx] = vq - c.marginal;
                    out.afd_key[slot * out.afd_capacity + idx] = lkey_r;
                }
            }
            // the other samples' lists: only operand sets that equal the MAP in the integrated sample as well
            unsigned long long todo = __ballot(on && !dup && xq == mx);
            while (todo) {
                const int qq = q0 + __builtin_ctzll(todo);
                todo &= todo - 1;
                afd_consider(c, uni_d(V[qq]), sin_r, uni_d(X[qq]), bulk ? sin_r : -1);
            }
        }
    }
    afd_finish(c);
}

// ------------------------------------------------------------------------------------------------
// Two builds of the same kernel: WPE = 2 waves per SIMD (no spills) for workgroups whose LDS footprint allows only 8 of
// them per CU anyway, WPE = 3 (168 VGPRs, 32 of them spilled) where 9 or more fit (single-sample 30x: 12 workgroups,
// +33 %).  The launcher picks by LDS bytes.
#ifdef VLR_DBG_NUM_VGPR  // diagnosis builds: a VGPR cap independent of the launch bounds (run at VLR_WAVES_PER_SIMD=2)
#define VLR_DBG_VGPR_ATTR __attribute__((amdgpu_num_vgpr(VLR_DBG_NUM_VGPR)))
#else
#define VLR_DBG_VGPR_ATTR
#endif
constexpr int kGridQuantum = kXcds;   // the launchers round the grid up to a multiple of this (XCD-aware locus mapping below)
template <int WPE>
__global__ void __launch_bounds__(64, WPE) VLR_DBG_VGPR_ATTR vlr_call_kernel(const DevPlan plan_arg, DevBatch batch, DevResults out,
                                                           int max_obs, int range_depth) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ WaveSt wst;
    const DevPlan& p = plan_arg;
    const int lane = threadIdx.x;
    // XCD-aware mapping: consecutive workgroup ids go round the eight XCDs (each with its own L2), so workgroup w takes locus
    // (w mod 8) * ceil(n / 8) + w / 8 — every XCD works through ONE contiguous eighth of the batch, and the loci whose column
    // slices share cache lines at their ends (4 B x ~100 observations per column and pileup: three to four 128 B lines, two of them
    // shared with the neighbours) meet in the same L2 at about the same time instead of being filled into two L2s.  Measured on
    // config 3 (profiles/r06g.md): L2 fills + write-backs 12.49 -> 10.00 GB per million loci (1.55 -> 1.24 x the algorithmic bytes) for
    // + 0.4 % kernel time; plans with more than two samples lose 1.6 - 2.8 % (configs 4, 5) and keep w -> w.  Tiled variants (8 to
    // 1024 consecutive loci per XCD in turn) were slower than both.  VLR_NO_XCD_MAP: the A/B build.
    // (The grid is rounded up to a multiple of eight: kXcds * ceil(n / 8) ids cover 0 .. n - 1 exactly once.)
#ifdef VLR_NO_XCD_MAP
    const int64_t locus = blockIdx.x;
#else
    const int64_t locus = plan_arg.S <= kXcdMapMaxSamples
                              ? (int64_t)(blockIdx.x % kXcds) * ((batch.n_loci + kXcds - 1) / kXcds) + (int64_t)(blockIdx.x / kXcds)
                              : (int64_t)blockIdx.x;
#endif
    if (locus >= batch.n_loci) return;   // (plans above kLdsSamples samples: the host launches the wide build)
    if (VLR_DEEP && !(out.status[locus] & VLR_LOCUS_TOO_DEEP)) return;  // deep launch: only what the LDS-resident kernel could not hold
    // replay launch next to an AFD log: only the loci whose log region overflowed are re-evaluated
    if (!VLR_DEEP && out.replay && out.afd_log && __double_as_longlong(out.afd_log[(size_t)locus * (size_t)out.afd_log_stride]) >= 0) return;
    const int S = p.S;
    WaveSt* w = &wst;

    Ctx c;
    c.plan = &plan_arg; c.w = w; c.lane = lane; c.S = S;
    const int cap = p.table_cap;
    c.cap = cap;
    c.coef = dyn;
    c.ecoef = out.escratch + (size_t)locus * (size_t)(max_obs + 2 * p.S) + 2 * p.S;  // (the first 2 S words of the row: ones_ptr)
    c.tabX = dyn + 2 * max_obs;
    c.tabV = c.tabX + p.max_tab_depth * cap;
    c.rowX = c.tabV + p.max_tab_depth * cap;
    c.rowV = c.rowX + kRows * cap;
    // sort scratch of integrate_table aliases row 1: it is only used by outer chains (whose inner chains are
    // finished) and by the single-chain fallback (which owns row 0)
    c.sx = c.rowX + cap;
    c.sv = c.rowV + cap;
    c.tvaf = c.rowV + kRows * cap;
    double* evM = c.tvaf + kRows * S;
    double* evS = evM + p.n_univ;
    const int n_slots = p.n_univ + 1;       // + virtual artifact slot of the `absent` group
    double* mapJ = evS + p.n_univ;          // [n_slots]
    double* mapVaf = mapJ + n_slots;        // [n_slots][S]
    c.setv = mapVaf + n_slots * S;          // [S][kMaxSet]
    c.cacheA = c.setv + S * p.max_set;      // [S][kCacheWays] x 3   (setv: [S][max_set], the plan's largest Set spectrum)
    c.cacheB = c.cacheA + S * kCacheWays;
    c.cacheV = c.cacheB + S * kCacheWays;
    c.afd_seen = c.cacheV + S * kCacheWays;  // [S][seen_cap] x (VAF, l2fc key)
    c.seen_cap = p.max_set > 16 ? p.max_set : 16;
    const int n_seen = out.replay ? 2 * S * c.seen_cap + 2 * S : 0;  // only the AFD replay pass records discrete VAFs
    c.mapv = c.afd_seen + 2 * S * c.seen_cap;
    c.afd_nseen = (int*)(c.mapv + S);
    int* mapHyp = (int*)(c.afd_seen + n_seen);  // [n_slots]
    c.dkeyV = c.afd_seen + n_seen + (n_slots + 1) / 2 + 2;  // [n_dkey]
    c.nframes = p.max_frames; c.nrs = range_depth;
    c.frames = (Frame*)(c.dkeyV + p.n_dkey);
    c.rs = (RangeSt*)(c.frames + c.nframes);
    c.mapJ = mapJ; c.mapVaf = mapVaf; c.mapHyp = mapHyp; c.n_slots = n_slots;
    c.status = 0;
    if (lane == 0) { w->work[0] = 0; w->work[1] = 0; }
    c.need_batch = 0; c.bt_nt = 0; c.bt_inner = 0;
    c.replay = out.replay; c.locus = locus; c.outp = &out; c.mapGroup = 0; c.mapDisc = 0; c.marginal = 0.0; c.afd_cnt = nullptr;
    c.lg = (out.afd_log && !out.replay && !VLR_DEEP) ? out.afd_log + (size_t)locus * (size_t)out.afd_log_stride : nullptr;
    c.lg_pos = kLogFirst; c.lg_cap = (int)out.afd_log_stride; c.lg_nrec = 0; c.hyp = 0;
#ifdef VLR_PROFILE
    for (int i = 0; i < 40; ++i) c.prof[i] = 0;
#endif
    PROF_START(c);

    const unsigned lf = batch.locus_flags[locus];
    c.vt = batch.variant_type ? batch.variant_type[locus] : 0;
    if (c.vt >= kNVariantTypes) c.vt = VLR_VT_OTHER;
    c.has_snv = (lf & VLR_LOCUS_HAS_SNV) != 0;
    c.refbase = batch.ref_base ? batch.ref_base[locus] : 0;
    c.altbase = batch.alt_base ? batch.alt_base[locus] : 0;
    const bool remove_nonstd = (lf & VLR_LOCUS_REMOVE_NONSTANDARD) != 0;
    const unsigned bias_mask = lf & 0x3f;

    // ============================ phase A: pileup statistics ============================
    // (preprocess_record's pileup edits calling.rs:581-625; gating inputs bias/*.rs; all predicates use
    //  the ORIGINAL prob_alt/prob_ref, read_observation.rs:425-452)
    int n_alt_like = 0, total_kept = 0, filtered = 0;
    int n_uncertain = 0, strong_ref_std = 0, strong_ref_f1r2 = 0;        // read_orientation_bias.rs:38-97
    int n_alt_s = 0, nm_alt = 0, n_ref_s = 0, nm_ref = 0;                 // alt_locus_bias.rs:124-144
    bool any_altloci = false, any_softclip = false;
    unsigned possible = 0;                                                 // bit h: some obs is bias evidence under h
    unsigned possible_alb_noloci = 0;
    int offset_acc = 0;
    bool too_deep = false;
    double mx_sb_all = VLR_NEG_INF, mx_sb_fwd = VLR_NEG_INF;
    int ehas_mask = 0;
    for (int s = 0; s < S; ++s) {
        const int64_t pidx = locus * S + s;
        const uint32_t o0 = batch.obs_offset[pidx], o1 = batch.obs_offset[pidx + 1];
        int nk = 0, allref = 1, allpos = 1, sall = 0, anysa = 0, ins = 0, del = 0;
        int sbias[kNHyp];
        int sbias_alb_noloci = 0;
        double mx_all = VLR_NEG_INF, mx_major = VLR_NEG_INF, mx_rate = VLR_NEG_INF;
        for (int h = 0; h < kNHyp; ++h) sbias[h] = 0;
        // all columns this pass needs in one round of loads (none waits for another), and the rows of the NEXT 64 observations are
        // requested before this iteration's arithmetic starts
        struct RowA { float pm, pa, pr, psa, phb; uint32_t f; };
        auto load_a = [&](uint32_t i) {
            RowA r{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u};
            if (i < o1) { r.pm = batch.pm[i]; r.pa = batch.pa[i]; r.pr = batch.pr[i]; r.f = batch.flags[i]; r.psa = batch.psa[i]; r.phb = batch.phb[i]; }
            return r;
        };
        RowA nxa = load_a(o0 + lane);
        for (uint32_t base = o0; base < o1; base += 64) {
            uint32_t i = base + lane;
            bool valid = i < o1;
            const RowA cua = nxa;
            if (base + 64 < o1) nxa = load_a(i + 64);
            double pm = cua.pm, pa = cua.pa, pr = cua.pr;
            float psa_f = cua.psa, phb_f = cua.phb;
            uint32_t f = cua.f;
            bool keep = valid && !(remove_nonstd && f_orient(f) == VLR_ORIENT_OTHER);  // pileup.rs:26-43
            filtered += popc64(__ballot(valid && !keep));
            if (__ballot(keep && psa_f != 0.0f)) ehas_mask |= 1 << s;  // s = e^psa != 1: third coefficient e != 0
            // Bayes factors against their thresholds in log space: pr - pa is a difference of two f32 values, i.e. a multiple of
            // 2^-22 or coarser near ln 20, so it never comes within an ulp of the threshold and exp(d) > 20 <=> d > ln 20 holds for
            // every possible input (NaN and the infinities compare the same way) — no exponential needed
            const double dra = pr - pa;
            bool strong_ref = keep && dra > kLn20;         // read_observation.rs:434-437 (KassRaftery >= Strong)
            bool strong_alt = keep && -dra > kLn20;        // 429-432
            bool pos_ref = dra > kLn3;                     // 443-446
            bool ref_sup = pr > pa;                        // 439-441
            bool uniq = pm >= kLn095;                      // 425-427
            nk += popc64(__ballot(keep));
            n_alt_like += popc64(__ballot(keep && pa > pr));
            if (__ballot(keep && !ref_sup)) allref = 0;
            if (__ballot(keep && !pos_ref)) allpos = 0;
            bool sa_u = strong_alt && uniq;
            sall += popc64(__ballot(sa_u));
            if (__ballot(strong_alt)) anysa = 1;
            int hl = f_hplen(f);
            if (__ballot(keep && hl > 0)) ins = 1;
            if (__ballot(keep && hl < 0)) del = 1;
            int orient = f_orient(f);
            bool std_or = orient == VLR_ORIENT_F1R2 || orient == VLR_ORIENT_F2R1;
            n_uncertain += popc64(__ballot(keep && !std_or));
            strong_ref_std += popc64(__ballot(strong_ref && std_or));
            strong_ref_f1r2 += popc64(__ballot(strong_ref && orient == VLR_ORIENT_F1R2));
            bool maxq = (f & VLR_F_MAX_MAPQ) != 0;
            n_alt_s += popc64(__ballot(strong_alt));
            nm_alt += popc64(__ballot(strong_alt && !maxq));
            n_ref_s += popc64(__ballot(strong_ref));
            nm_ref += popc64(__ballot(strong_ref && !maxq));
            if (__ballot(keep && f_altlocus(f) != VLR_ALTLOCUS_NONE)) any_altloci = true;
            if (__ballot(keep && (f & VLR_F_SOFTCLIPPED))) any_softclip = true;
            {   // maxima of the decision sums (second pass below)
                const double phb1 = (double)phb_f;
                const int st1 = f_strand(f);
                mx_sb_all = fmax(mx_sb_all, (strong_ref && st1 != VLR_STRAND_BOTH) ? pm : VLR_NEG_INF);
                mx_sb_fwd = fmax(mx_sb_fwd, (strong_ref && st1 == VLR_STRAND_FORWARD) ? pm : VLR_NEG_INF);
                mx_all = fmax(mx_all, strong_ref ? pm : VLR_NEG_INF);
                mx_major = fmax(mx_major, (strong_ref && (f & VLR_F_READPOS_MAJOR)) ? pm : VLR_NEG_INF);
                mx_rate = fmax(mx_rate, strong_ref ? pm + phb1 : VLR_NEG_INF);
            }
            for (int h = 1; h < kNHyp; ++h) {
                if (h == H_HE) continue;
                bool ev = bias_evidence(h, f, true);
                if (__ballot(keep && ev)) possible |= (1u << h);
                sbias[h] += popc64(__ballot(sa_u && ev));
            }
            {
                bool ev = bias_evidence(H_ALB, f, false);
                if (__ballot(keep && ev)) possible_alb_noloci = 1;
                sbias_alb_noloci += popc64(__ballot(sa_u && ev));
            }
        }
        if (lane == 0) {
            w->nkeep[s] = nk; w->soff[s] = offset_acc;
            w->all_ref[s] = allref; w->all_posref[s] = allpos ? 1 : 0; w->strong_all[s] = sall;
            w->any_strong_alt[s] = anysa; w->has_ins[s] = ins; w->has_del[s] = del;
            for (int h = 0; h < kNHyp; ++h) w->strong_bias[s][h] = sbias[h];
            w->strong_bias[s][0] = sbias_alb_noloci;  // slot 0 reused: alt-locus evidence without alt loci
        }
        {
            const double a0 = wave_max(mx_all), a1 = wave_max(mx_major), a2 = wave_max(mx_rate);
            if (lane == 0) { w->pos_all[s] = a0; w->pos_major[s] = a1; w->pos_rate[s] = a2; }  // maxima; replaced by the sums below
        }
        offset_acc += nk;
        total_kept += nk;
    }
    // second pass (rows are L2-hot): the exp(ln_sum_exp(prob_mapping)) sums of strand_bias.rs:79-123 and
    // read_position_bias.rs:63-122, kept apart from the counters above to bound register pressure
    const double m_sb_all = uni_d(wave_max(mx_sb_all)), m_sb_fwd = uni_d(wave_max(mx_sb_fwd));
    DdAcc sb_all{{0.0, 0.0}}, sb_fwd{{0.0, 0.0}};
    VLR_SYNC();
    for (int s = 0; s < S; ++s) {
        const int64_t pidx = locus * S + s;
        const uint32_t o0 = batch.obs_offset[pidx], o1 = batch.obs_offset[pidx + 1];
        const double m_all = uni_d(w->pos_all[s]), m_major = uni_d(w->pos_major[s]), m_rate = uni_d(w->pos_rate[s]);
        DdAcc pa_all{{0.0, 0.0}}, pa_major{{0.0, 0.0}}, pa_rate{{0.0, 0.0}};
        struct RowB { float pm, pa, pr, phb; uint32_t f; };
        auto load_b = [&](uint32_t i) {
            RowB r{0.0f, 0.0f, 0.0f, 0.0f, 0u};
            if (i < o1) { r.pm = batch.pm[i]; r.pa = batch.pa[i]; r.pr = batch.pr[i]; r.phb = batch.phb[i]; r.f = batch.flags[i]; }
            return r;
        };
        RowB nxb = load_b(o0 + lane);
        for (uint32_t base = o0; base < o1; base += 64) {
            uint32_t i = base + lane;
            bool valid = i < o1;
            const RowB cub = nxb;
            if (base + 64 < o1) nxb = load_b(i + 64);
            double pm = cub.pm, pa = cub.pa, pr = cub.pr, phb = cub.phb;
            uint32_t f = cub.f;
            bool keep = valid && !(remove_nonstd && f_orient(f) == VLR_ORIENT_OTHER);
            bool strong_ref = keep && (pr - pa) > kLn20;
            int strand = f_strand(f);
            // exp(pm - maximum) once per distinct maximum: the four sums over prob_mapping differ in their subsets only, and their
            // maxima coincide whenever the reads share the top mapping quality (same argument, same value: bit-identical sums)
            const bool on_pm = strong_ref && pm != VLR_NEG_INF;
            double e_all = 0.0;
            if (__ballot(on_pm)) e_all = vlr_det::det_exp(pm - m_all);
            ddacc_add_e(sb_all, (m_sb_all == m_all) ? e_all : vlr_det::det_exp(pm - m_sb_all), on_pm && strand != VLR_STRAND_BOTH);
            ddacc_add_e(sb_fwd, (m_sb_fwd == m_all) ? e_all : vlr_det::det_exp(pm - m_sb_fwd), on_pm && strand == VLR_STRAND_FORWARD);
            ddacc_add_e(pa_all, e_all, on_pm);
            ddacc_add_e(pa_major, (m_major == m_all) ? e_all : vlr_det::det_exp(pm - m_major), on_pm && (f & VLR_F_READPOS_MAJOR));
            ddacc_add(pa_rate, pm + phb, m_rate, strong_ref);
        }
        const double e_all = ddacc_exp(pa_all, m_all), e_major = ddacc_exp(pa_major, m_major), e_rate = ddacc_exp(pa_rate, m_rate);
        VLR_SYNC();
        if (lane == 0) { w->pos_all[s] = e_all; w->pos_major[s] = e_major; w->pos_rate[s] = e_rate; }
    }
#if VLR_DEEP
    int obs_cap = 0;
    {   // {c, q} pairs and e of every kept observation from the pool: 3 doubles per observation
        const unsigned long long need = 3ull * (unsigned long long)offset_acc + 2ull * (unsigned long long)S;  // (+ the all-ones words in front of e)
        unsigned long long at = 0;
        if (lane == 0) at = atomicAdd(out.deep_used, need);
        at = (unsigned long long)UNI64((long long)at);
        if (at + need > (unsigned long long)out.deep_capacity) too_deep = true;
        else {
            c.coef = out.deep_pool + at;
            c.ecoef = out.deep_pool + at + 2ull * (unsigned long long)offset_acc + 2ull * (unsigned long long)S;
            obs_cap = offset_acc;
        }
    }
#else
    if (offset_acc > max_obs) too_deep = true;
    // the bump counter of the deep launch that follows this one in stream order (workgroup 0 resets it: no memset node per launch)
    if (blockIdx.x == 0 && lane == 0 && out.deep_used) *out.deep_used = 0ull;
#endif
    if (lane == 0) w->ehas = ehas_mask;
    c.ehas = ehas_mask;
    VLR_SYNC();
    if (filtered > 0) c.status |= VLR_LOCUS_FILTERED_ALN;
    if (total_kept == 0) c.status |= VLR_LOCUS_MISSING_DATA;
    const bool singleton = (n_alt_like == 1);  // adjust_singleton_evidence (read_observation.rs:548-562)
    if (singleton) c.status |= VLR_LOCUS_SINGLETON_ADJ;

    PROF_ADD(c, 0);  // phase A statistics
    // ---- learn_parameters (bias/mod.rs:295-300)
    double forward_rate = 0.5;
    bool sb_informative = false;
    {
        double strong_all_f = ddacc_exp(sb_all, m_sb_all), strong_fwd_f = ddacc_exp(sb_fwd, m_sb_fwd);
        if (strong_all_f > 2.0) {
            double ff = strong_fwd_f / strong_all_f;
            if (strong_all_f > 100.0 && ff > 0.0 && ff < 1.0) { forward_rate = ff; sb_informative = true; }
            else if (ff >= 0.4 && ff <= 0.6) { forward_rate = 0.5; sb_informative = true; }
        }
    }
    forward_rate = uni_d(forward_rate);
    const bool has_alt_loci = any_altloci;

    // ---- gating (modes/generic.rs:443-448; bias/mod.rs:232-257)
    unsigned enabled = 0;  // hypotheses in the event's bias list (bias/mod.rs:131-218)
    if (bias_mask & VLR_BIAS_ALTLOCUS) enabled |= 1u << H_ALB;
    if (bias_mask & VLR_BIAS_HOMOPOLYMER) enabled |= 1u << H_HE;
    if (bias_mask & VLR_BIAS_SOFTCLIP) enabled |= 1u << H_SCB;
    if (bias_mask & VLR_BIAS_POSITION) enabled |= 1u << H_RPB;
    if (bias_mask & VLR_BIAS_ORIENTATION) enabled |= (1u << H_F1R2) | (1u << H_F2R1);
    if (bias_mask & VLR_BIAS_STRAND) enabled |= (1u << H_SBF) | (1u << H_SBR);
    const int n_biases = __popc(enabled);
    unsigned surviving = 0;
    if (!too_deep) {
        bool he_inf = true;  // homopolymer_error.rs:46-72
        for (int s = 0; s < S; ++s) he_inf = he_inf && (!w->any_strong_alt[s] || (w->has_ins[s] && w->has_del[s]));
        bool rob_inf;
        {
            bool enough = (double)n_uncertain < ((double)total_kept / 2.0);
            bool uniform = false;
            if (strong_ref_std > 2) {
                double fr = (double)strong_ref_f1r2 / (double)strong_ref_std;
                uniform = fr >= 0.3 && fr <= 0.7;
            }
            rob_inf = enough && uniform;
        }
        bool rpb_inf = false;  // read_position_bias.rs:63-122
        for (int s = 0; s < S; ++s) {
            double ea = w->pos_all[s];
            if (ea > 10.0) {
                double em = w->pos_major[s], er = w->pos_rate[s];
                double major_rate = em / ea;
                if (em > 0.0 && fabs(major_rate - er) < 0.05) rpb_inf = true;
            }
        }
        bool alb_inf;
        {
            bool enough_alt = n_alt_s > 0 && (double)nm_alt > ((double)n_alt_s * 0.1) && (n_alt_s - nm_alt) < 10;
            bool enough_ref = n_ref_s > 0 && ((double)nm_ref < ((double)n_ref_s * 0.9));
            alb_inf = enough_alt && (has_alt_loci || enough_ref);
        }
        for (int h = 1; h < kNHyp; ++h) {
            if (!(enabled & (1u << h))) continue;
            bool ok;
            if (h == H_HE) ok = he_inf;  // is_possible = is_likely = is_informative (homopolymer_error.rs:74-80)
            else {
                bool poss = (h == H_ALB && !has_alt_loci) ? (possible_alb_noloci != 0) : ((possible >> h) & 1u);
                bool inf = (h == H_ALB) ? alb_inf : (h == H_SCB) ? any_softclip : (h == H_RPB) ? rpb_inf
                           : (h == H_F1R2 || h == H_F2R1) ? rob_inf : sb_informative;
                bool likely = false;  // bias/mod.rs:60-104
                for (int s = 0; s < S; ++s) {
                    int sa = w->strong_all[s];
                    int sbv = (h == H_ALB && !has_alt_loci) ? w->strong_bias[s][0] : w->strong_bias[s][h];
                    bool r;
                    if (sa >= 10) r = ((double)sbv / (double)sa) >= 0.66666;
                    else if (w->all_ref[s]) r = false;
                    else if (w->nkeep[s] == 0) r = false;
                    else r = true;
                    likely = likely || r;
                }
                ok = poss && inf && likely;
            }
            if (ok) surviving |= (1u << h);
        }
    }

    TRC(c, 1, total_kept); TRC(c, 2, n_alt_like); TRC(c, 3, forward_rate); TRC(c, 4, (int)surviving); TRC(c, 5, (int)enabled); TRC(c, 6, offset_acc);
    PROF_ADD(c, 1);  // gating
    // ============================ phase B: hypotheses x events ============================
    for (int u = lane; u < p.n_univ; u += 64) { evM[u] = VLR_NEG_INF; evS[u] = 0.0; }
    for (int u = lane; u < n_slots; u += 64) { mapJ[u] = VLR_NEG_INF; mapHyp[u] = -1; }
    VLR_SYNC();

    if (too_deep) c.status |= VLR_LOCUS_TOO_DEEP;
    unsigned hyps = too_deep ? 0u : (1u | surviving);
    if (c.replay) {
        // AFD only exists for a non-artifact MAP (calling.rs:889): MAP operands, best event group and marginal come
        // from the first pass; only the clean events are re-evaluated
        bool have = true;
        for (int s = 0; s < S; ++s) { double v = out.map_vaf[locus * S + s]; if (v != v) have = false; }
        bool art = false;
        if (out.map_bias) for (int i = 0; i < VLR_N_BIAS; ++i) art = art || out.map_bias[locus * VLR_N_BIAS + i] != 0;
        if (!have || art || too_deep) {
            if (VLR_DEEP && !too_deep && lane == 0) out.status[locus] &= ~VLR_LOCUS_TOO_DEEP;  // no lists to make: done
            return;
        }
        VLR_SYNC();
        if (lane < S) { c.mapv[lane] = out.map_vaf[locus * S + lane]; c.afd_nseen[lane] = 0; }
        VLR_SYNC();
        int be = UNI(out.best_event[locus]);
        c.mapGroup = (be == 0) ? 0 : ((be - 1) / 2 + 1);
        c.mapDisc = UNI((int)out.map_disc[locus]);
        c.marginal = uni_d(out.ln_marginal[locus]);
        hyps = 1u;
    }
    // wave-uniform doubles of the hypothesis loop, computed once and parked in SGPRs (left to the compiler they are hoisted
    // into VGPRs and spilled to scratch across the loop)
    const double ln_bias_share = uni_d(kLn05 + log(1.0 / (double)n_biases));
    const SgprD reverse_rate = park_sd(1.0 - forward_rate);
    for (int h = 0; h < kNHyp; ++h) {
        if (!((hyps >> h) & 1u)) continue;
        c.hyp = h;
        TRC(c, 10, h);
        // ---- per-observation affine coefficients for this hypothesis -> LDS
        // L_i(alpha, beta) = c_i + q_i*alpha + e_i*beta with
        //   w = e^pm, u = (1-w) * e^(missed + b_any), A = e^(pa + b_alt), R = e^(pr + b_ref), s = e^prob_sample_alt
        //   c = w*R + u, q = w*s*(A-R), e = w*(1-s)*(A-R)
        // (likelihood.rs:43-53,86-115,171-220; bias factors bias/mod.rs:259-284)
        int fastmask = 0, vfastmask = 0, onesmask = 0;
        for (int s = 0; s < S; ++s) {
            const int64_t pidx = locus * S + s;
            const uint32_t o0 = batch.obs_offset[pidx], o1 = batch.obs_offset[pidx + 1];
            int wr = w->soff[s];
            int fast_s = 1, vfast_s = 1;
            const bool ehas_s = (ehas_mask >> s) & 1;
            // Underflow rescue: a term that is finite in the reference's log space but zero or subnormal in linear space
            // (|log-prob| beyond ~708) makes the pass run a second time for this sample with every observation's coefficients
            // scaled by its own power of two 2^-k_i; the exponents add up to kshift[s], which every pileup log-likelihood of the
            // sample gets back (kshift_ln).  Never taken on pair-HMM output (supports are normalised, realignment/mod.rs:359-374).
            bool need_rescue = false;
            int kacc = 0;
            // the terms of one kept observation under hypothesis h: the affine coefficients, the term at alpha = beta = 1 formed directly
            // (one_t) and what c + q + e cancels there (ref_t); `scaled`: relative to the observation's own power of two 2^ki
            auto obs_terms = [&](const ObsRow& cur, int scaled, double& cc_, double& cq_, double& ce_, double& one_t, double& ref_t, bool& uflow, int& ki) {
                const uint32_t f = cur.f;
                ki = 0;
                    double pm = cur.pm, pa = cur.pa, pr = cur.pr, miss = cur.miss;
                    double psa = cur.psa, pdo = cur.pdo, phb = cur.phb;
                    double hpa = cur.hpa;
                    double hpv = cur.hpv;
                    if (singleton && pa > pr) { pa = kLn05; pr = kLn05; }  // prob_alt_adj / prob_ref_adj
                    int strand = f_strand(f), orient = f_orient(f);
                    bool major = (f & VLR_F_READPOS_MAJOR) != 0;
                    // read position: prob_any (read_position_bias.rs:27-37,51-62)
                    double one_minus_hit = (phb != 0.0) ? -expm1(phb) : 1.0;
                    double rp_any = major ? exp(phb) : one_minus_hit;
                    // linear-space bias factors for alt / ref / any
                    double fa, fr, fany;
                    // strand (strand_bias.rs:30-58)
                    double sb_alt;
                    if (h == H_SBF) sb_alt = (strand == VLR_STRAND_FORWARD || strand == VLR_STRAND_NONE) ? 1.0 : 0.0;
                    else if (h == H_SBR) sb_alt = (strand == VLR_STRAND_REVERSE || strand == VLR_STRAND_NONE) ? 1.0 : 0.0;
                    else if (strand == VLR_STRAND_BOTH) sb_alt = exp(pdo);
                    else if (strand == VLR_STRAND_NONE) sb_alt = 1.0;
                    else {
                        double rate = (strand == VLR_STRAND_FORWARD) ? forward_rate : fresh_sd(reverse_rate);
                        sb_alt = rate * (-expm1(pdo));  // ln(rate) + prob_single_overlap
                    }
                    // orientation (read_orientation_bias.rs:18-36)
                    double ro_alt = 0.5;
                    if (h == H_F1R2) ro_alt = (orient == VLR_ORIENT_F1R2) ? 1.0 : (orient == VLR_ORIENT_F2R1) ? 0.0 : 0.5;
                    else if (h == H_F2R1) ro_alt = (orient == VLR_ORIENT_F2R1) ? 1.0 : (orient == VLR_ORIENT_F1R2) ? 0.0 : 0.5;
                    // position (read_position_bias.rs:18-25)
                    double rp_alt = (h == H_RPB) ? (major ? 1.0 : 0.0) : rp_any;
                    // softclip (softclip_bias.rs:15-29)
                    double sc_alt = (h == H_SCB) ? ((f & VLR_F_SOFTCLIPPED) ? 1.0 : 0.0) : 1.0;
                    // homopolymer (homopolymer_error.rs:23-44): prob_ref = prob_alt, prob_any = 1
                    double he_lp = (h == H_HE) ? hpa : hpv;
                    double he_alt = 1.0;
                    if (__ballot(he_lp == he_lp)) he_alt = (he_lp == he_lp) ? exp(he_lp) : 1.0;  // (no homopolymer evidence in the whole row: no exponential)
                    // alt locus (alt_locus_bias.rs:63-113)
                    double al_alt = 0.5, al_ref = 0.5;
                    if (h == H_ALB) {
                        if (has_alt_loci) {
                            bool mj = f_altlocus(f) == VLR_ALTLOCUS_MAJOR;
                            al_alt = mj ? 1.0 : 0.0;
                            al_ref = mj ? 0.0 : 1.0;
                        } else {
                            al_alt = (f & VLR_F_MAX_MAPQ) ? 0.0 : 1.0;
                            al_ref = 0.5;
                        }
                    }
                    fa = sb_alt * ro_alt * rp_alt * sc_alt * he_alt * al_alt;
                    fr = 0.5 * 0.5 * rp_any * 1.0 * he_alt * al_ref;
                    fany = 0.5 * 0.5 * rp_any * 0.5;
                    double wv = exp(pm);
                    double mis = -expm1(pm);  // prob_mismapping = ln_one_minus_exp(pm) (read_observation.rs:283-286)
                    double A = exp(pa) * fa, R = exp(pr) * fr;
                    double uu = mis * exp(miss) * fany;
                    const double sv = ehas_s ? exp(psa) : 1.0;  // (!ehas: prob_sample_alt == 0 on every kept observation of the sample, e^0 = 1)
                    uflow = (A == 0.0 && fa != 0.0 && pa > VLR_NEG_INF) || (R == 0.0 && fr != 0.0 && pr > VLR_NEG_INF);
                    double d = A - R;
                    cc_ = wv * R + uu; cq_ = wv * sv * d; ce_ = wv * (1.0 - sv) * d;
                    one_t = wv * A + uu; ref_t = wv * R;  // the term at alpha = beta = 1 formed directly, and what c + q + e cancels
                    TRC(c, 130, sb_alt); TRC(c, 131, fa); TRC(c, 132, fr); TRC(c, 133, A); TRC(c, 134, R); TRC(c, 135, wv); TRC(c, 136, uu); TRC(c, 137, sv); TRC(c, 138, d); TRC(c, 139, cq_);
                    TRC(c, 140, exp(pa)); TRC(c, 141, rp_any); TRC(c, 142, he_alt); TRC(c, 143, ro_alt); TRC(c, 144, forward_rate); TRC(c, 145, fresh_sd(reverse_rate)); TRC(c, 146, pa); TRC(c, 147, strand);
                    if (__builtin_expect(scaled != 0, 0)) {
                        // the three log-space addends of the observation's likelihood, their largest as the binary exponent k
                        const double lA = (fa > 0.0 && pa > VLR_NEG_INF) ? pm + pa + log(fa) : VLR_NEG_INF;
                        const double lR = (fr > 0.0 && pr > VLR_NEG_INF) ? pm + pr + log(fr) : VLR_NEG_INF;
                        const double lU = (mis > 0.0 && fany > 0.0 && miss > VLR_NEG_INF) ? log(mis) + miss + log(fany) : VLR_NEG_INF;
                        const double mx = fmax(lA, fmax(lR, lU));
                        ki = (mx > VLR_NEG_INF) ? (int)floor(mx / kLn2) : 0;
                        const double sh = (double)ki * kLn2;
                        const double WA = (lA > VLR_NEG_INF) ? exp(lA - sh) : 0.0, WR = (lR > VLR_NEG_INF) ? exp(lR - sh) : 0.0;
                        const double UU = (lU > VLR_NEG_INF) ? exp(lU - sh) : 0.0;
                        d = WA - WR;
                        cc_ = WR + UU; cq_ = sv * d; ce_ = (1.0 - sv) * d;
                        one_t = WA + UU; ref_t = WR;
                        uflow = false;
                    }
            };
          for (int scaled = 0; scaled < 2; ++scaled) {
            if (scaled && !need_rescue) break;
            wr = w->soff[s]; fast_s = 1; vfast_s = 1;
            bool fatal = false;
            bool risk = false;    // some term has w R > 2^24 (w A + u): c + q + e keeps fewer than 29 bits of it at alpha = beta = 1
            // every column of a row in one round of loads (the few rows that are dropped below are loaded in vain), and the rows of
            // the NEXT 64 observations are requested before this iteration's arithmetic starts
            ObsRow nxt = load_obs_row(batch, o0 + lane, o1);
            for (uint32_t base = o0; base < o1; base += 64) {
                uint32_t i = base + lane;
                bool valid = i < o1;
                bool tiny = false, small = false;
                const ObsRow cur = nxt;
                if (base + 64 < o1) nxt = load_obs_row(batch, i + 64, o1);
                const uint32_t f = cur.f;
                bool keep = valid && !(remove_nonstd && f_orient(f) == VLR_ORIENT_OTHER);
                unsigned long long km = __ballot(keep);
                int pos = wr + popc64(km & ((1ull << lane) - 1ull));
                if (keep) {
                    double cc_, cq_, ce_, one_t, ref_t;
                    bool uflow;
                    int ki;
                    obs_terms(cur, scaled, cc_, cq_, ce_, one_t, ref_t, uflow, ki);
                    kacc += ki;
                    risk = risk || (ref_t > 0x1p24 * one_t);
#if VLR_DEEP
                    if (pos < obs_cap) {
#else
                    if (pos < max_obs) {
#endif
                        c.coef[2 * pos + 0] = cc_;
                        c.coef[2 * pos + 1] = cq_;
                        if (ehas_s) __hip_atomic_store(const_cast<double*>(c.ecoef) + pos, ce_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    // smallest value the term can take over alpha in [0,1], beta in [0,1] (linear => at a corner)
                    double mn = fmin(fmin(cc_, cc_ + cq_), fmin(cc_ + ce_, cc_ + cq_ + ce_));
                    tiny = !(mn >= 0x1p-200);  // also catches NaN
                    // e^x of a finite log-probability came out as zero: harmless while the term itself stays a normal number
                    // (whatever underflowed is below 5e-324, i.e. < 2.5e-16 of any normal term: the reference's ln_add_exp
                    // drops it too).  Only a term that is itself unrepresentable makes the locus unusable in linear space.
                    if (uflow && !(mn >= 0x1p-1022)) fatal = true;
                    small = !(mn >= 0x1p-70 && fmax(fmax(cc_, cc_ + cq_), fmax(cc_ + ce_, cc_ + cq_ + ce_)) <= 2.0);
                }
                if (__ballot(tiny)) fast_s = 0;
                if (__ballot(small)) vfast_s = 0;
                wr += popc64(km);
            }
            const bool bad = __ballot(fatal) != 0ull;
            if (!scaled) need_rescue = bad;
            else if (bad) c.status |= VLR_LOCUS_UNDERFLOW;  // cannot happen: the largest addend of every term is in [1/2, 1)
            if (!(bad && !scaled)) {  // (the pass that stands)
                if (__builtin_expect(__ballot(risk) != 0ull, 0)) {
                    // rare (never on pair-HMM records): one more pass over the sample's rows for prod_i (w A_i + u_i), the likelihood
                    // at alpha = beta = 1 without the cancellation — mantissa and exponent, lane shares combined below
                    double Pw[1] = {1.0};
                    int Ew[1] = {0};
                    for (uint32_t base = o0; base < o1; base += 64) {
                        const ObsRow cur = load_obs_row(batch, base + lane, o1);
                        if (base + lane < o1 && !(remove_nonstd && f_orient(cur.f) == VLR_ORIENT_OTHER)) {
                            double cc_, cq_, ce_, one_t, ref_t;
                            bool uflow;
                            int ki, e1;
                            obs_terms(cur, scaled, cc_, cq_, ce_, one_t, ref_t, uflow, ki);
                            Pw[0] *= __builtin_frexp(one_t, &e1);
                            Ew[0] += e1;
                            Pw[0] = __builtin_frexp(Pw[0], &e1);
                            Ew[0] += e1;
                        }
                    }
                    reduce_terms<1, 64>(Pw, Ew);
                    if (lane == 0) {
                        __hip_atomic_store(ones_ptr(c) + 2 * s, Pw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(ones_ptr(c) + 2 * s + 1, (double)Ew[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    onesmask |= 1 << s;
                }
            }
          }
            {
                const int ks = need_rescue ? (int)wave_sum((double)kacc) : 0;
                if (lane == 0) kshift(c)[s] = ks;
            }
            if (fast_s) fastmask |= 1 << s;
            if (vfast_s) vfastmask |= 1 << s;
        }
        if (lane == 0) { w->fastok = fastmask; w->vfast = vfastmask; }
        c.ehas = ehas_mask | (onesmask << 16);
        if (lane < S) w->cacheN[lane] = 0;
        VLR_SYNC();  // also orders the e coefficients (HBM scratch row, written by other lanes than the ones that read them)
        if (__ballot((c.status & VLR_LOCUS_UNDERFLOW) != 0)) c.status |= VLR_LOCUS_UNDERFLOW;
        if (p.n_dkey > 0 && !c.replay) {  // pileup likelihoods of the flattened discrete roots under this hypothesis
            for (int k = 0; k < p.n_dkey; ++k) {
                const double r = sample_lik_point(c, ldc(&p.dkey[k].sample), ldc(&p.dkey[k].a), ldc(&p.dkey[k].b));
                if (lane == 0) c.dkeyV[k] = r;
            }
            VLR_SYNC();
        }

        TRC(c, 11, c.coef[2 * (lane < offset_acc ? lane : 0)]); TRC(c, 12, c.coef[2 * (lane < offset_acc ? lane : 0) + 1]);
        TRC(c, 13, w->fastok); TRC(c, 14, w->vfast); TRC(c, 15, c.ehas); TRC(c, 16, kshift(c)[lane < S ? lane : 0]);
        PROF_ADD(c, 2);  // coefficient pass
        // ---- events (calling.rs:654-687): absent + clean events under h = none, artifact twins otherwise
        const double bias_prior = (h == 0) ? kLn05 : ln_bias_share;  // modes/generic.rs:437-441
        const int first_ev = (h == 0) ? -1 : 0;
        // pass 0 (probe): roots that are a single innermost chain are deferred and run together, one per DPP row;
        // pass 1: the remaining (nested / branching / set-valued) roots through the general walk
        c.ndef = 0;
        c.deferred = 0;
        c.nhold = 0; c.nstash = 0; c.hold_inner = -1;
        unsigned long long todo = 0ull;  // roots left for pass 1 (bit = running root counter, first 64 roots)
        // The two passes over (event, root) are an explicit iterator so that the chain batches — asked for by a resumable
        // walk (outer Range over a leaf Range) or by a full / final set of deferred event-level chains — all run at ONE
        // inlined run_chain_batch site (three copies of that loop cost code size, registers and instruction-cache hits).
        {
            enum { IT_NEXT, IT_ROOT, IT_WALK, IT_DONE };
            int pass = 0, e = first_ev, rc_ = 0, st = IT_NEXT;
            int ri = (e < 0) ? 0 : ldc(p.root_off + e);
            int r1 = (e < 0) ? 1 : ldc(p.root_off + e + 1);
            int u = 0, root = 0, resume = 0;
            bool fresh = true;  // (e, ri) not yet considered
            for (;;) {
                // run_kind 1: batch of the walk (+ held chains riding along in its free rows), 2: a full set of deferred chains, then
                // the same root again, 3: the held chains that found no batch to ride along with, at the end of pass 1
                int run_mask = 0, run_inner = 0, run_kind = 0, piggy = 0;
                if (st == IT_NEXT) {
                    // advance to the next root this pass has to look at
                    bool found = false;
                    for (;;) {
                        if (!fresh) { ++ri; ++rc_; }
                        fresh = false;
                        while (ri >= r1) {
                            ++e;
                            if (e >= p.n_named) break;
                            ri = ldc(p.root_off + e);
                            r1 = ldc(p.root_off + e + 1);
                        }
                        if (e >= p.n_named) break;
                        const bool probe = (pass == 0) && rc_ < 64;
                        if (pass == 0 && !probe) continue;
                        if (pass == 1 && rc_ < 64 && !((todo >> rc_) & 1ull)) continue;
                        found = true;
                        break;
                    }
                    if (!found) {
                        if (pass == 0) {  // end of the probe pass: run what was deferred, then the general pass
                            pass = 1; e = first_ev; rc_ = 0; fresh = true;
                            ri = (e < 0) ? 0 : ldc(p.root_off + e);
                            r1 = (e < 0) ? 1 : ldc(p.root_off + e + 1);
                            // (a FULL set of four cannot be held — two rows and the stash take three —: it runs as a batch of its own here,
                            //  as it would have if a fifth deferrable root had followed)
                            if (c.ndef > 0 && (c.replay || todo == 0ull || c.ndef == kRows)) {  // nothing left that could give them a ride
                                run_kind = 3; run_mask = (1 << c.ndef) - 1; run_inner = UNI(w->task[0].inner); c.ndef = 0;
                            } else if (c.ndef > 0) {
                                // Held chains: instead of a batch of their own (three of four rows busy in the tumor-normal
                                // scenarios) the deferred chains ride along in the free rows of the batches that the nested
                                // roots of pass 1 ask for (2, 3, 3, ... of four rows busy).  Two are parked in the top rows,
                                // a third in WaveSt::stash; bo_setup leaves the parked rows alone.
                                const int nd = c.ndef;
                                c.hold_inner = UNI(w->task[0].inner);
                                move_task(c, nd - 1, kRows - 1);
                                if (nd >= 2) move_task(c, nd - 2, kRows - 2);
                                if (nd >= 3) move_task(c, nd - 3, -1);
                                c.nhold = nd >= 2 ? 2 : 1; c.nstash = nd >= 3 ? 1 : 0; c.ndef = 0;
                            }
                        } else if (c.nhold + c.nstash > 0) {  // nobody gave them a ride
                            if (c.nstash) { move_task(c, -1, kRows - 1 - c.nhold); c.nhold += 1; c.nstash = 0; }
                            run_kind = 3; run_mask = ((1 << c.nhold) - 1) << (kRows - c.nhold); run_inner = c.hold_inner;
                            c.nhold = 0;
                        } else st = IT_DONE;
                    } else if (c.ndef == kRows) { run_kind = 2; run_mask = (1 << kRows) - 1; run_inner = UNI(w->task[0].inner); c.ndef = 0; }
                    else st = IT_ROOT;
                }
                PROF_ADD(c, 19);  // iterator: next root
                if (st == IT_DONE) break;
                if (st == IT_ROOT) {
                    u = (e < 0) ? 0 : (1 + 2 * e + (h == 0 ? 0 : 1));
                    c.group = e + 1;
                    c.defer_ok = (1 - pass) & (int)((unsigned)(rc_ - 64) >> 31);  // pass == 0 && rc_ < 64, as integer arithmetic (stays a scalar)
                    c.defer_slot = u;
                    const DevFastRoot* fr = p.froot + ((e < 0) ? 0 : 1 + ri);
                    const bool is_droot = !(c.replay || p.n_dkey == 0) && ldc(p.droot + 2 * ((e < 0) ? 0 : 1 + ri)) >= 0;  // all-discrete roots: below
#ifdef VLR_NO_FAST_CODE
                    const int fkind = 0;
#else
                    const int fkind = (c.defer_ok && !is_droot) ? ldc(&fr->kind) : 0;
#endif
                    if (fkind != 0) {  // compiled chain root: the task straight from the plan's record; 2: certainly a root for the general pass
                        const int fk = fkind == 1 ? fast_chain_root(c, fr) : 2;
                        if (fk == 2) todo |= 1ull << rc_;
                        st = IT_NEXT;
                        PROF_ADD(c, 34);
                        continue;
                    }
                    VLR_SYNC();
                    c.curJ = uni_d(mapJ[u]);
                    c.curHyp = UNI(mapHyp[u]);
                    if (lane < S) w->curMapVaf[lane] = mapVaf[u * S + lane];
                    VLR_SYNC();
                    root = (e < 0) ? p.absent_root : ldc(p.roots + ri);
                    const int di = (e < 0) ? 0 : 1 + ri;
                    const int dl0 = (c.replay || p.n_dkey == 0) ? -1 : ldc(p.droot + 2 * di);
                    if (dl0 >= 0) {  // all-discrete root: its leaves side by side on the lanes
                        const double dens = eval_discrete_root(c, dl0, ldc(p.droot + 2 * di + 1));
                        PROF_ADD(c, 23);
                        if (dens != dens) c.status |= VLR_LOCUS_NAN;
                        double M = uni_d(evM[u]), Sx = uni_d(evS[u]);
                        lse_add(M, Sx, bias_prior + dens);
                        VLR_SYNC();
                        if (lane == 0) { evM[u] = M; evS[u] = Sx; mapJ[u] = c.curJ; mapHyp[u] = c.curHyp; }
                        if (lane < S) mapVaf[u * S + lane] = w->curMapVaf[lane];
                        VLR_SYNC();
                        st = IT_NEXT;
                    } else { resume = 0; st = IT_WALK; }
                    PROF_ADD(c, 20);  // root entry (slot state, discrete roots are counted apart)
                }
                if (st == IT_WALK) {
                    const double dens = uni_d(walk_root(c, root, resume));
                    TRC(c, 24, dens); TRC(c, 25, u); TRC(c, 26, c.need_batch); TRC(c, 27, c.deferred);
                    PROF_ADD(c, 21);  // walk (descent, frames; the outer-batch steps are counted apart)
                    if (c.need_batch) {
                        c.need_batch = 0; run_kind = 1; run_mask = (1 << c.bt_nt) - 1; run_inner = c.bt_inner;
                        if ((c.nhold | c.nstash) && c.hold_inner == c.bt_inner) {  // (batches of the walk carry no l2fc terms)
                            if (c.nhold == 0 && c.bt_nt < kRows) { move_task(c, -1, kRows - 1); c.nhold = 1; c.nstash = 0; }
                            if (c.nhold) { piggy = ((1 << c.nhold) - 1) << (kRows - c.nhold); run_mask |= piggy; }
                        }
                    }
                    else {
                        if (c.deferred == 1) c.deferred = 0;                                 // delivered by flush_deliver
                        else if (c.deferred == 2) { c.deferred = 0; todo |= 1ull << rc_; }  // second pass
                        else {
                            if (dens != dens) c.status |= VLR_LOCUS_NAN;
                            double M = uni_d(evM[u]), Sx = uni_d(evS[u]);
                            lse_add(M, Sx, bias_prior + dens);
                            VLR_SYNC();
                            if (lane == 0) { evM[u] = M; evS[u] = Sx; mapJ[u] = c.curJ; mapHyp[u] = c.curHyp; }
                            if (lane < S) mapVaf[u * S + lane] = w->curMapVaf[lane];
                            VLR_SYNC();
                        }
                        st = IT_NEXT;
                    }
                }
                PROF_ADD(c, 22);  // root exit (event accumulators, slot state)
                if (run_kind) {
                    PROF_ADD(c, 31);  // event loop: between the walk's return and the batch
                    if (run_kind != 1) c.nlfc = 0;  // deferred chains carry no l2fc terms; a later probe walk may have left some in the context
                    TRC(c, 20, run_mask); TRC(c, 21, run_inner); TRC(c, 22, run_kind); TRC(c, 23, piggy);
                    run_chain_batch(c, run_mask, run_inner);
                    VLR_SYNC();
                    if (run_kind == 1) {
                        if (piggy) {
                            // the walk is suspended with the MAP candidate of ITS slot and its operands in the context: park them
                            // in the slot arrays, hand the held chains to their events, and take the context back
                            const int sg = c.group, sd = c.disc, sc = c.contained, sn = c.nlfc;
                            const alive_t sa = c.alive;
                            const double so = w->ops_vaf[lane < S ? lane : 0];
                            VLR_SYNC();
                            if (lane == 0) { mapJ[u] = c.curJ; mapHyp[u] = c.curHyp; }
                            if (lane < S) mapVaf[u * S + lane] = w->curMapVaf[lane];
                            VLR_SYNC();
                            flush_deliver(c, piggy, evM, evS, bias_prior);
                            c.curJ = uni_d(mapJ[u]); c.curHyp = UNI(mapHyp[u]);
                            if (lane < S) { w->curMapVaf[lane] = mapVaf[u * S + lane]; w->ops_vaf[lane] = so; }
                            VLR_SYNC();
                            c.group = sg; c.disc = sd; c.contained = sc; c.alive = sa; c.nlfc = sn;
                            c.nhold = 0;
                        }
                        resume = 1; st = IT_WALK;
                    } else {
                        flush_deliver(c, run_mask, evM, evS, bias_prior);
                        st = (run_kind == 2) ? IT_ROOT : IT_NEXT;
                    }
                    PROF_ADD(c, 4);  // delivery of deferred / held chains
                }
            }
        }
        PROF_ADD(c, 3);
    }
    PROF_ADD(c, 3);  // walk remainder (everything in the event loop not attributed below)
    if (c.replay) {
        afd_finish(c);
        if (VLR_DEEP && lane == 0) out.status[locus] &= ~VLR_LOCUS_TOO_DEEP;  // evaluated by the deep call launch, lists done
        return;
    }
    // ============================ phase C: posteriors + MAP ============================
    // bio Model::compute: marginal = ln_sum_exp(event values); posterior = value - marginal
    const int n_out = p.n_named + 2;
    // the value of event slot u (M + ln S of its streaming sum) once, on lane u (n_univ <= 2 kMaxNamedEvents + 1 = 61): the loops
    // below used to take the same logarithm four times per slot on all lanes
    VLR_SYNC();
#ifdef VLR_WIDE_BUILD
    // wide build: up to 2 x 62 + 1 = 125 slots, more than one per lane: the values go through the LDS (evS[u] becomes the slot's value)
    for (int u = lane; u < p.n_univ; u += 64) evS[u] = lse_value(evM[u], evS[u]);
    VLR_SYNC();
#define EV_M(u) evM[u]
#define EV_V(u) evS[u]
#else
    const int u_l = lane < p.n_univ ? lane : 0;
    const double evM_l = evM[u_l];
    const double evV_l = lse_value(evM_l, evS[u_l]);
    TRC(c, 110, evM_l); TRC(c, 111, evV_l); TRC(c, 112, evS[u_l]);
#define EV_M(u) lane_d(evM_l, u)
#define EV_V(u) lane_d(evV_l, u)
#endif
    double mM = VLR_NEG_INF, mS = 0.0;
    for (int u = 0; u < p.n_univ; ++u) {
        const double m_u = EV_M(u);
        double v = (m_u != m_u) ? m_u : EV_V(u);
        lse_add(mM, mS, v);
    }
    double marginal = (mM != mM) ? mM : lse_value(mM, mS);
    TRC(c, 113, marginal);
    if (marginal != marginal) c.status |= VLR_LOCUS_NAN;
    // call_record (calling.rs:762-803)
    int best = 0;
    double best_post = VLR_NEG_INF;
    double aM = VLR_NEG_INF, aS = 0.0;
    bool have_twins = n_biases > 0;
    for (int u = 0; u < p.n_univ; ++u) {
        bool twin = (u > 0) && ((u & 1) == 0);
        if (twin && !have_twins) continue;
        double v = EV_V(u);
        double post = v - marginal;
        if (u == 0 || !(post < best_post)) { best = u; best_post = post; }  // last maximum wins (itertools minmax)
        if (twin) lse_add(aM, aS, post);
    }
    double prob_artifact = lse_value(aM, aS);
    bool is_artifact = true;
    for (int u = 0; u < p.n_univ; ++u) {
        bool twin = (u > 0) && ((u & 1) == 0);
        if (twin) continue;
        double post = EV_V(u) - marginal;
        if (!(post < prob_artifact)) is_artifact = false;
    }
    double* lp = out.ln_posterior + locus * n_out;
    {   // posterior of `absent` (slot 0) and of the clean named events (slots 1 + 2 e), each written by the lane that holds the slot's value:
        // no lane read inside the `lane == 0` region below (a lane read of lanes that are off there — round 6, EXEC assert build)
#ifdef VLR_WIDE_BUILD
        for (int u = lane; u < p.n_univ; u += 64)
            if (u == 0 || (u & 1)) lp[u == 0 ? 0 : 1 + (u >> 1)] = evS[u] - marginal;
#else
        const double post_l = evV_l - marginal;
        if (lane < p.n_univ && (lane == 0 || (lane & 1))) lp[lane == 0 ? 0 : 1 + (lane >> 1)] = post_l;
#endif
    }
#undef EV_M
#undef EV_V
    if (lane == 0) {
        lp[n_out - 1] = prob_artifact;
        if (out.ln_marginal) out.ln_marginal[locus] = marginal;
        if (out.best_event) out.best_event[locus] = best;
    }
    // sample_infos (calling.rs:844-937): MAP among operands of the best event's tree (clean + twin share it)
    {
        VLR_SYNC();
        int uc = (best == 0) ? 0 : (((best - 1) / 2) * 2 + 1);
        int ua = (best == 0) ? p.n_univ : uc + 1;
        int pick = -1;
        if (mapHyp[uc] >= 0) pick = uc;
        if (is_artifact && have_twins && mapHyp[ua] >= 0) {
            if (pick < 0) pick = ua;
            else {
                // same comparator as map_consider: prob desc, then lower hypothesis id (clean = 0 wins ties)
                if (mapJ[ua] > mapJ[uc]) pick = ua;
            }
        }
        if (lane < S) {
            double v = __builtin_nan("");
            if (pick >= 0) v = ((mapHyp[pick] & 15) > 0) ? 0.0 : mapVaf[pick * S + lane];
            out.map_vaf[locus * S + lane] = v;
        }
        if (out.map_bias && lane == 0) {
            uint8_t* mb = out.map_bias + locus * VLR_N_BIAS;
            for (int i = 0; i < VLR_N_BIAS; ++i) mb[i] = 0;
            int hh = pick >= 0 ? (mapHyp[pick] & 15) : 0;
            if (out.map_disc) out.map_disc[locus] = (uint16_t)(pick >= 0 ? (mapHyp[pick] >> 4) : 0);
            switch (hh) {
                case H_SBF: mb[0] = 1; break;
                case H_SBR: mb[0] = 2; break;
                case H_F1R2: mb[1] = 1; break;
                case H_F2R1: mb[1] = 2; break;
                case H_RPB: mb[2] = 1; break;
                case H_SCB: mb[3] = 1; break;
                case H_HE: mb[4] = 1; break;
                case H_ALB: mb[5] = 1; break;
                default: break;
            }
        }
    }
    if (c.lg && lane == 0) {  // words used (-1: overflow, the replay launch takes over) and the record count
        c.lg[0] = __longlong_as_double((long long)c.lg_pos);
        c.lg[1] = __longlong_as_double((long long)c.lg_nrec);
    }
    // deep launch with AFD lists: the deep replay launch still has to find this locus
    if (VLR_DEEP && out.afd_count && !(c.status & VLR_LOCUS_TOO_DEEP)) c.status |= VLR_LOCUS_TOO_DEEP;
    if (lane == 0) {
        out.status[locus] = c.status;
        if (out.work) {
            atomicAdd(&out.work[0], w->work[0]);
            atomicAdd(&out.work[1], w->work[1]);
#ifdef VLR_PROFILE
            PROF_ADD(c, 9);  // phase C
            for (int i = 0; i < 40; ++i) atomicAdd(&out.work[2 + i], c.prof[i]);
#endif
        }
    }
}

}  // namespace vlr

// ---- diagnostics: the device build of the decision arithmetic (include/vlr_detmath.h) and of ln_mantissa, element-wise
namespace vlr {
__global__ void vlr_selftest_math_kernel(int which, const double* a, const double* b, double* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (which) {
        case 0: r = vlr_det::det_exp(a[i]); break;
        case 1: r = vlr_det::det_log1p_pos(a[i]); break;
        case 2: r = vlr_det::det_log2_ratio(a[i], b[i]); break;
        case 3: r = vlr_det::det_exp2(a[i]); break;
        default: r = ln_mantissa(a[i]); break;
    }
    out[i] = r;
}
}  // namespace vlr
// ---- diagnostics: streams of known size in the engine's own access widths, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// (the microarchitecture guide calibrates them for 16 B/lane reads only): mode 0 reads n f32 with lane-contiguous 4-byte
// loads (the observation columns), mode 1 writes n f64 with lane-contiguous 8-byte stores (the e scratch, the results)
namespace vlr {
__global__ void __launch_bounds__(64) vlr_selftest_stream_kernel(const float* in, double* out, long long n, int mode) {
    const long long per_wg = 64 * 256;  // a wave streams 256 consecutive 256-byte segments, like a deep pileup
    const long long base = (long long)blockIdx.x * per_wg;
    if (mode == 0) {
        float acc = 0.f;
        for (int k = 0; k < 256; ++k) {
            const long long i = base + (long long)k * 64 + threadIdx.x;
            if (i < n) acc += in[i];
        }
        if (acc == 12345.678f) out[blockIdx.x] = acc;  // keep the loads alive without writing
    } else {
        for (int k = 0; k < 256; ++k) {
            const long long i = base + (long long)k * 64 + threadIdx.x;
            if (i < n) out[i] = (double)i;
        }
    }
}
}  // namespace vlr
// Wide build (vlr_kernels_wide.hip: VLR_WIDE_BUILD, VLR_LDS_SAMPLES 16, namespace renamed): the same kernels with per-sample LDS
// arrays for sixteen samples; exports vlr_launch_call_kernel_wide and vlr_launch_afd_kernel_wide and nothing else.
#ifdef VLR_WIDE_BUILD
#define VLR_FN_CALL vlr_launch_call_kernel_wide
#define VLR_FN_AFD vlr_launch_afd_kernel_wide
#define VLR_FN_LDS vlr_plan_lds_floor_wide
#else
#define VLR_FN_CALL vlr_launch_call_kernel
#define VLR_FN_AFD vlr_launch_afd_kernel
#define VLR_FN_LDS vlr_plan_lds_floor
#endif
#ifdef VLR_DBG_OWNER  // host side of the diagnosis builds (VLR_DBG_EXEC_ASSERT / VLR_DBG_TRACE above); not in a shipped library
#ifdef VLR_DBG_EXEC_ASSERT
// out[4 * line + {0: executed, 1: executions under partial EXEC, 2: executions breaking the site's rule, 3: OR of disabled lanes}]
extern "C" int vlr_debug_exec_sites(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vlr::g_xsite), sizeof(unsigned long long) * vlr::kXSites * 4) != hipSuccess) return 2;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(vlr::g_xsite)) != hipSuccess) return 3;
        if (hipMemset(p, 0, sizeof(unsigned long long) * vlr::kXSites * 4) != hipSuccess) return 4;
    }
    return 0;
}
#endif
#ifdef VLR_DBG_TRACE
extern "C" int vlr_debug_trace_arm(long long locus) {   // trace the wave of this locus of the following launches; -1: none
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vlr::g_trace_n), &zero, sizeof(int)) != hipSuccess) return 2;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vlr::g_trace_locus), &locus, sizeof(long long)) != hipSuccess) return 3;
    return 0;
}
// records written since vlr_debug_trace_arm (at most cap): hdr[2 r] = (id << 32) | source line, hdr[2 r + 1] = EXEC, val[64 r + lane]
extern "C" long long vlr_debug_trace_read(unsigned long long* hdr, double* val, long long cap) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(vlr::g_trace_n), sizeof(int)) != hipSuccess) return -2;
    long long k = n < vlr::kTraceCap ? n : vlr::kTraceCap;
    if (k > cap) k = cap;
    if (k > 0) {
        if (hipMemcpyFromSymbol(hdr, HIP_SYMBOL(vlr::g_trace_hdr), sizeof(unsigned long long) * 2 * (size_t)k) != hipSuccess) return -3;
        if (hipMemcpyFromSymbol(val, HIP_SYMBOL(vlr::g_trace_val), sizeof(double) * 64 * (size_t)k) != hipSuccess) return -4;
    }
    return (long long)n;
}
#endif
#endif
#if !VLR_DEEP && !defined(VLR_WIDE_BUILD)
extern "C" int vlr_launch_selftest_stream(const float* in, double* out, long long n, int mode, void* stream) {
    if (n <= 0) return 0;
    const long long per_wg = 64 * 256;
    hipLaunchKernelGGL(vlr::vlr_selftest_stream_kernel, dim3((unsigned)((n + per_wg - 1) / per_wg)), dim3(64), 0, (hipStream_t)stream, in, out, n, mode);
    return (int)hipGetLastError();
}

extern "C" int vlr_launch_selftest_math(int which, const double* a, const double* b, double* out, long long n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(vlr::vlr_selftest_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, which, a, b, out, n);
    return (int)hipGetLastError();
}

#endif  // !VLR_DEEP && !VLR_WIDE_BUILD (selftest launchers)
#if VLR_DEEP
// deep launcher: the 2-waves-per-SIMD instance only (no coefficient area in LDS; max_obs = 0 for the layout)
#ifdef VLR_WIDE_BUILD
#define VLR_FN_DEEP vlr_launch_call_kernel_widedeep
#else
#define VLR_FN_DEEP vlr_launch_call_kernel_deep
#endif
extern "C" int VLR_FN_DEEP(const vlr::DevPlan* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out,
                           int n_univ, int n_samples, int range_depth, void* stream) {
    using namespace vlr;
    if (batch->n_loci <= 0) return 0;
    if (n_samples > kLdsSamples) return (int)hipErrorInvalidValue;  // (the per-sample LDS arrays of this build; the host picks the wide build)
    if (range_depth < 1) range_depth = 1;
    size_t n_slots = (size_t)n_univ + 1;
    size_t cap = (size_t)plan_host->table_cap;
    size_t dbl = (size_t)2 * plan_host->max_tab_depth * cap + (size_t)2 * kRows * cap + (size_t)kRows * n_samples +
                 (size_t)2 * n_univ + n_slots + n_slots * n_samples + (size_t)n_samples * plan_host->max_set + (size_t)(out->replay ? 2 * n_samples * (plan_host->max_set > 16 ? plan_host->max_set : 16) + 2 * n_samples : 0) + (size_t)3 * n_samples * kCacheWays +
                 (n_slots + 1) / 2 + 2 + (size_t)plan_host->n_dkey +
                 ((size_t)plan_host->max_frames * sizeof(Frame) + (size_t)range_depth * sizeof(RangeSt) + 7) / 8 +
                 (size_t)(n_samples + 1) / 2;
    size_t bytes = dbl * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*)vlr_call_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(vlr_call_kernel<2>, dim3((unsigned)(((batch->n_loci + kGridQuantum - 1) / kGridQuantum) * kGridQuantum)), dim3(64), bytes, (hipStream_t)stream, *plan_host, *batch, *out, 0, range_depth);
    return (int)hipGetLastError();
}
#else
extern "C" int VLR_FN_AFD(const vlr::DevPlan* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out, void* stream) {
    if (batch->n_loci <= 0) return 0;
    const size_t seen_bytes = (size_t)2 * plan_host->S * (plan_host->max_set > 16 ? plan_host->max_set : 16) * sizeof(double);
    if (seen_bytes > 32768) {  // (large Set spectra: above the default dynamic LDS limit; vlr_plan_create has checked that it fits the CU)
        const hipError_t e = hipFuncSetAttribute((const void*)vlr::vlr_afd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)seen_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(vlr::vlr_afd_kernel, dim3((unsigned)batch->n_loci), dim3(64), seen_bytes, (hipStream_t)stream, *plan_host, *batch, *out);
    return (int)hipGetLastError();
}

// dynamic LDS of one workgroup of the call kernel (the layout at the top of vlr_call_kernel)
static size_t call_kernel_dyn_lds(const vlr::DevPlan* plan_host, int n_univ, int n_samples, int max_obs, int range_depth, bool replay) {
    using namespace vlr;
    size_t n_slots = (size_t)n_univ + 1;
    size_t cap = (size_t)plan_host->table_cap;
    size_t dbl = (size_t)2 * max_obs + (size_t)2 * plan_host->max_tab_depth * cap + (size_t)2 * kRows * cap + (size_t)kRows * n_samples +
                 (size_t)2 * n_univ + n_slots + n_slots * n_samples + (size_t)n_samples * plan_host->max_set + (size_t)(replay ? 2 * n_samples * (plan_host->max_set > 16 ? plan_host->max_set : 16) + 2 * n_samples : 0) + (size_t)3 * n_samples * kCacheWays +
                 (n_slots + 1) / 2 + 2 + (size_t)plan_host->n_dkey +
                 ((size_t)plan_host->max_frames * sizeof(Frame) + (size_t)range_depth * sizeof(RangeSt) + 7) / 8 +
                 (size_t)(n_samples + 1) / 2;  // kshift[S] (ints)
    return dbl * sizeof(double);
}
// Worst-case LDS of a plan's launches without any coefficient area (max_obs = 0): call pass, AFD replay and the AFD log filter.  A plan
// whose tables alone do not fit the 160 KiB of a CU (Set spectra of hundreds of members in many samples) is refused by vlr_plan_create
// with this number instead of failing at its first batch with a launch error.
extern "C" long long VLR_FN_LDS(const vlr::DevPlan* plan_host, int n_univ, int n_samples, int range_depth) {
    if (range_depth < 1) range_depth = 1;
    const size_t call = call_kernel_dyn_lds(plan_host, n_univ, n_samples, 0, range_depth, true) + 4096;   // + the static part (WaveSt, < 4 KiB in every build)
    const size_t afd = (size_t)2 * plan_host->S * (plan_host->max_set > 16 ? plan_host->max_set : 16) * sizeof(double) + 16384;  // + static part of vlr_afd_kernel
    return (long long)(call > afd ? call : afd);
}

// LDS bytes of one workgroup of the call kernel (static + dynamic) at a pileup budget of max_obs: what the launcher's rule below sees
// (vlr_plan_fit_max_obs in vlr_host.cpp sizes the budget of shallow batches with it)
#ifndef VLR_WIDE_BUILD
extern "C" long long vlr_launch_call_lds_bytes(const vlr::DevPlan* plan_host, int n_univ, int n_samples, int max_obs, int range_depth) {
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, (const void*)vlr::vlr_call_kernel<2>) != hipSuccess) return -1;
    if (range_depth < 1) range_depth = 1;
    return (long long)(fa.sharedSizeBytes + call_kernel_dyn_lds(plan_host, n_univ, n_samples, max_obs, range_depth, false));
}
#endif
// host-callable launcher (used by vlr_host.cpp)
extern "C" int VLR_FN_CALL(const vlr::DevPlan* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out,
                                      int n_univ, int n_samples, int max_obs, int range_depth, void* stream) {
    using namespace vlr;
    if (batch->n_loci <= 0) return 0;
    if (n_samples > kLdsSamples) return (int)hipErrorInvalidValue;  // (the per-sample LDS arrays of this build; the host picks the wide build)
    if (range_depth < 1) range_depth = 1;
    const size_t bytes = call_kernel_dyn_lds(plan_host, n_univ, n_samples, max_obs, range_depth, out->replay != 0);
    static size_t static_lds = 0;
    if (!static_lds) {
        hipFuncAttributes fa{};
        hipError_t ea = hipFuncGetAttributes(&fa, (const void*)vlr_call_kernel<2>);
        if (ea != hipSuccess) return (int)ea;
        static_lds = fa.sharedSizeBytes ? fa.sharedSizeBytes : 1;
    }
    // 160 kB of LDS per CU: the 2-wave build tops out at 8 workgroups per CU; from 9 on the 3-wave build wins although it
    // spills (tools/occupancy_probe.py: +5 % at 9-10 workgroups, +15..33 % at 10-12, -4 % at 8)
    // Round 6 (tools/waves4_probe.py, waves4_lds_probe.py): where SIXTEEN workgroups fit a CU, the 4-wave build (128 VGPRs, ~85 of them
    // spilled) wins on the shallow multi-sample workloads — tumor-normal at 20x / 30x: +11 % / +13 % — and is level on config 2 (+1 %); at
    // 14-15 workgroups it is +2 %, below that it only pays its spills (-9 %).  (Getting config 3 there — 2 kB of row tables moved to the
    // HBM scratch row plus a pileup budget at the 99.7 % quantile — was built and measured: +1.7 % on config 3, +2.3 % on config 4,
    // -4 % on config 5; not kept.  profiles/r06c.md)
    const size_t lds_wg = (static_lds + bytes + 511) & ~(size_t)511;
    int wpe = (static_lds + bytes <= (size_t)kLdsWg16) ? 4 : (163840 / lds_wg >= 9) ? 3 : 2;
    // single-sample plans (one chain at a time, no nested frames: the walk's state is small) with room for 20 workgroups or more: the
    // 6-wave build (80 VGPRs).  config 2: 5.71 -> 5.35 ms per 200 000 loci (+6.7 %); a two-sample plan at the same footprint loses half
    // (tumor-normal at 15x: 26.8 -> 41.1 ms), 8 waves lose everywhere (tools/waves8_probe.py)
    if (n_samples == 1 && 163840 / lds_wg >= 20) wpe = 6;
    if (const char* ev = getenv("VLR_WAVES_PER_SIMD")) wpe = atoi(ev);  // tuning / build-matrix knob (tests/test_gpu_build_matrix.py)
    if (getenv("VLR_DEBUG_LAUNCH")) fprintf(stderr, "vlr launch: max_obs %d static %zu dynamic %zu -> %zu B, %d waves/SIMD\n", max_obs, static_lds, bytes, static_lds + bytes, wpe);
    dim3 grid((unsigned)(((batch->n_loci + kGridQuantum - 1) / kGridQuantum) * kGridQuantum)), block(64);   // (a multiple of eight: the kernel's XCD-aware mapping)
#define VLR_LAUNCH(W)                                                                                                        \
    case W: {                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)vlr_call_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        if (e != hipSuccess) return (int)e;                                                                                  \
        hipLaunchKernelGGL(vlr_call_kernel<W>, grid, block, bytes, (hipStream_t)stream, *plan_host, *batch, *out, max_obs, range_depth); \
        break;                                                                                                               \
    }
    switch (wpe) {
        VLR_LAUNCH(6)
        VLR_LAUNCH(4)
        VLR_LAUNCH(3)
#ifdef VLR_STRESS_BUDGETS  // a register budget far below anything shipped: 64 VGPRs, hundreds of spills
        VLR_LAUNCH(8)
#endif
        default:
        VLR_LAUNCH(2)
    }
#undef VLR_LAUNCH
    return (int)hipGetLastError();
}
#endif  // VLR_DEEP / !VLR_DEEP launchers
