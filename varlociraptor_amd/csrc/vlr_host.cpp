// vlr_host.cpp — host side of the engine above the C ABI (include/vlr.h): plan compiler, prior
// tabulation, batch launch.  Compiled with hipcc into libvlr.so together with vlr_kernels.hip.
//
// Replaces, for the hot path only, Caller::configure_model (reference src/calling/variants/calling.rs:632-718),
// GenericModelBuilder::build (src/variants/model/modes/generic.rs:93-105) and the per-record
// Caller::call_record (calling.rs:720-842).  There is NO CPU fallback: without a HIP device every
// entry point that needs one fails with VLR_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vlr.h"
#include "vlr_plan.h"
#include "vlr_gpuio.h"

extern "C" int vlr_launch_afd_kernel(const vlr::DevPlanT<8>* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out, void* stream);
extern "C" int vlr_launch_call_kernel_deep(const vlr::DevPlanT<8>* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out,
                                           int n_univ, int n_samples, int range_depth, void* stream);
extern "C" int vlr_launch_call_kernel_widedeep(const vlr::DevPlanT<16>* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out,
                                               int n_univ, int n_samples, int range_depth, void* stream);
extern "C" int vlr_launch_afd_kernel_wide(const vlr::DevPlanT<16>* plan_host, const vlr::DevBatch* batch, const vlr::DevResults* out, void* stream);
extern "C" int vlr_launch_call_kernel_wide(const vlr::DevPlanT<16>* plan_dev, const vlr::DevBatch* batch, const vlr::DevResults* out,
                                           int n_univ, int n_samples, int max_obs, int range_depth, void* stream);
extern "C" int vlr_launch_call_kernel(const vlr::DevPlanT<8>* plan_dev, const vlr::DevBatch* batch, const vlr::DevResults* out,
                                      int n_univ, int n_samples, int max_obs, int range_depth, void* stream);
// LDS a workgroup of the plan needs before any coefficient area (call pass, AFD replay, AFD log filter; vlr_kernels.hip)
extern "C" long long vlr_plan_lds_floor(const vlr::DevPlanT<8>* plan_host, int n_univ, int n_samples, int range_depth);
extern "C" long long vlr_launch_call_lds_bytes(const vlr::DevPlanT<8>* plan_host, int n_univ, int n_samples, int max_obs, int range_depth);
extern "C" long long vlr_plan_lds_floor_wide(const vlr::DevPlanT<16>* plan_host, int n_univ, int n_samples, int range_depth);

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(VLR_ERR_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

const double NEG_INF = -std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------------------------------------
// Prior (reference src/variants/model/prior.rs), host restatement used ONLY to tabulate the prior over
// per-sample VAF classes (see build_prior_table).  bio LogProb helpers restated from the crate docs.
double ln_one_minus_exp(double p) { return p < -0.693 ? std::log1p(-std::exp(p)) : std::log(-std::expm1(p)); }
double ln_sum_exp(const std::vector<double>& v) {
    if (v.empty()) return NEG_INF;
    size_t im = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > v[im]) im = i;
    if (v[im] == NEG_INF) return NEG_INF;
    double s = 0;
    for (size_t i = 0; i < v.size(); ++i)
        if (i != im && v[i] != NEG_INF) s += std::exp(v[i] - v[im]);
    return v[im] + std::log1p(s);
}
bool relative_eq(double a, double b) {
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;
    double d = std::fabs(a - b), eps = std::numeric_limits<double>::epsilon();
    return d <= eps || d <= std::max(std::fabs(a), std::fabs(b)) * eps;
}

struct HostSpectrum {
    bool is_set;
    std::vector<double> set;
    double start, end;
    bool lex, rex;
    bool contains(double v) const {
        if (is_set) return std::find(set.begin(), set.end(), v) != set.end();
        bool lo = lex ? start < v : start <= v, hi = rex ? end > v : end >= v;
        return lo && hi;
    }
};

struct HostPrior {
    int S = 0;
    std::vector<uint8_t> uniform;
    std::vector<int> ploidy;
    std::vector<std::vector<HostSpectrum>> universe;
    std::vector<double> germline_rate, somatic_rate;
    std::vector<vlr_inheritance> inh;
    double het_ln = NAN;
    double var_het_ln = NAN, var_som_ln = NAN;  // per-variant overrides (prior.rs:250-270)
    double f_indel, f_mnv, f_sv;
    bool absent_only = true;
    int vt = VLR_VT_SNV;

    bool all_uniform() const { return std::all_of(uniform.begin(), uniform.end(), [](uint8_t u) { return u != 0; }); }  // prior.rs:111-113
    double vt_fraction() const {  // grammar/mod.rs:420-431
        return vt == VLR_VT_INDEL ? f_indel : vt == VLR_VT_MNV ? f_mnv : vt == VLR_VT_SV ? f_sv : 1.0;
    }
    bool som_ln(int s, double* o) const {  // prior.rs:250-257
        if (!std::isnan(var_som_ln)) { *o = var_som_ln; return true; }
        if (std::isnan(somatic_rate[s])) return false;
        *o = std::log(somatic_rate[s] * vt_fraction());
        return true;
    }
    bool het(double* o) const {  // prior.rs:263-270
        if (!std::isnan(var_het_ln)) { *o = var_het_ln; return true; }
        if (std::isnan(het_ln)) return false;
        *o = std::log(std::exp(het_ln) * vt_fraction());
        return true;
    }
    static double p_somatic(double rate, double v) { return relative_eq(v, 0.0) ? ln_one_minus_exp(rate) : rate; }  // 440-456
    using V = std::vector<double>;
    double clonal(int s, int par, const V& ev, const V& g, bool somatic) const {  // 458-512
        if (!relative_eq(g[s], g[par])) return NEG_INF;
        double r;
        bool has = som_ln(s, &r);
        if (somatic && has) return (ev[par] - g[par] != 0.0) ? 0.0 : p_somatic(r, ev[s] - g[s]);
        if (somatic) return relative_eq(ev[s] - g[s], ev[par] - g[par]) ? 0.0 : NEG_INF;
        if (has) return p_somatic(r, ev[s] - g[s]);
        return 0.0;
    }
    double subclonal(int s, int par, const V& ev, const V& g) const {  // 514-552
        if (!relative_eq(g[s], g[par])) return NEG_INF;
        double r;
        if (som_ln(s, &r)) return (ev[par] == 0.0 && g[s] == 0.0) ? p_somatic(r, ev[s]) : 0.0;
        return relative_eq(ev[s] - g[s], ev[par] - g[par]) ? 0.0 : NEG_INF;
    }
    double population(const std::vector<int>& pop, const V& g, double h) const {  // 554-582
        unsigned m = 0, n = 0;
        for (int s : pop) { m += (unsigned)std::llround(ploidy[s] * g[s]); n += (unsigned)ploidy[s]; }
        if (m > 0) return h - std::log((double)m);
        V v;
        for (unsigned i = 1; i <= n; ++i) v.push_back(h - std::log((double)i));
        return ln_one_minus_exp(ln_sum_exp(v));
    }
    static double binom(unsigned n, unsigned k) {
        if (k > n) return 0;
        double r = 1;
        for (unsigned i = 1; i <= k; ++i) r = r * (n - k + i) / i;
        return std::floor(0.5 + r);
    }
    static double hyper_ln(unsigned N, unsigned K, unsigned n, unsigned x) {  // 584-598 (statrs Hypergeometric::pmf)
        unsigned lo = n + K > N ? n + K - N : 0, hi = std::min(K, n);
        if (x < lo || x > hi) return NEG_INF;
        return std::log(binom(K, x) * binom(N - K, n - x) / binom(N, n));
    }
    bool mendel_counts(unsigned sp0, unsigned sp1, unsigned tp, unsigned sa0, unsigned sa1, unsigned ta, double gr, double* out) const {  // 600-678
        auto cases = [](unsigned p) {
            std::vector<unsigned> v;
            if (p % 2 == 0) v.push_back(p / 2);
            else { v.push_back(p / 2); v.push_back(p / 2 + 1); }
            return v;
        };
        bool valid = false;
        V probs;
        for (unsigned p1 : cases(sp0))
            for (unsigned p2 : cases(sp1)) {
                if (p1 + p2 != tp) continue;
                valid = true;
                for (unsigned a1 = 0; a1 <= std::min(sa0, p1); ++a1)
                    for (unsigned a2 = 0; a2 <= std::min(sa1, p2); ++a2) {
                        if (a1 + a2 > ta) continue;
                        double pr = hyper_ln(sp0, sa0, p1, a1) + hyper_ln(sp1, sa1, p2, a2);
                        probs.push_back(pr + std::log(gr) * (double)((int)ta - (int)(a1 + a2)));
                    }
            }
        if (!valid) return false;
        *out = ln_sum_exp(probs);
        return true;
    }
    bool mendelian(int c, int p0, int p1, const V& ev, const V& g, double* out) const {  // 680-712
        auto na = [&](int s) { return (unsigned)std::llround(g[s] * ploidy[s]); };
        double prob;
        if (!mendel_counts(ploidy[p0], ploidy[p1], ploidy[c], na(p0), na(p1), na(c), germline_rate[c] * vt_fraction(), &prob)) return false;
        double r;
        if (som_ln(c, &r)) prob += p_somatic(r, ev[c] - g[c]);
        *out = prob;
        return true;
    }
    // 298-438; `ok` is cleared when the reference would panic (ploidy mismatch)
    double calc(const V& ev, V g, bool* ok) const {
        if ((int)g.size() == S) {
            double prob = 0, h;
            if (het(&h)) {
                std::vector<int> pop;
                for (int s = 0; s < S; ++s)
                    if (inh[s].kind == VLR_INHERIT_NONE && ploidy[s] >= 0 && !uniform[s]) pop.push_back(s);
                prob = population(pop, g, h);
            }
            for (int s = 0; s < S; ++s) {
                if (uniform[s]) continue;
                double t = 0;
                switch (inh[s].kind) {
                    case VLR_INHERIT_MENDELIAN:
                        if (!mendelian(s, inh[s].from0, inh[s].from1, ev, g, &t)) { *ok = false; return NEG_INF; }
                        break;
                    case VLR_INHERIT_CLONAL: t = clonal(s, inh[s].from0, ev, g, inh[s].somatic != 0); break;
                    case VLR_INHERIT_SUBCLONAL: t = subclonal(s, inh[s].from0, ev, g); break;
                    default: {
                        double r;
                        if (som_ln(s, &r)) t = p_somatic(r, ev[s] - g[s]);
                    }
                }
                prob += t;
            }
            return prob;
        }
        int s = (int)g.size();
        auto push = [&](double x) { V g2 = g; g2.push_back(x); return g2; };
        if (ploidy[s] == 0 && ev[s] != 0.0) return NEG_INF;
        if (uniform[s]) {
            for (auto& sp : universe[s])
                if (sp.contains(ev[s])) return calc(ev, push(0.0), ok);
            return NEG_INF;
        }
        if (!std::isnan(somatic_rate[s])) {
            V probs;
            for (int n = 0; n <= ploidy[s]; ++n) probs.push_back(calc(ev, push(ploidy[s] > 0 ? (double)n / ploidy[s] : 0.0), ok));
            return ln_sum_exp(probs);
        }
        if (ploidy[s] >= 0 && !std::isnan(het_ln)) {
            double n_alt = ploidy[s] * ev[s];
            if (relative_eq(n_alt, std::round(n_alt))) return calc(ev, push(ev[s]), ok);
            return NEG_INF;
        }
        *ok = false;  // "bug: not enough info for prior but no universe specified"
        return NEG_INF;
    }
    double compute(const V& ev, bool* ok) const {  // 715-762
        bool absent = std::all_of(ev.begin(), ev.end(), [](double v) { return v == 0.0; });
        if (absent_only && !all_uniform()) {
            if (!absent) {
                double full = calc(ev, {}, ok);
                if (full == NEG_INF) return full;
                return ln_one_minus_exp(calc(V(S, 0.0), {}, ok));
            }
            return calc(ev, {}, ok);
        }
        return calc(ev, {}, ok);
    }
};

}  // namespace

struct vlr_plan {
    int device = 0;
    vlr::DevPlanT<16> host{};  // host copy (device pointers inside): the layout for up to sixteen samples
    long long lds_b0 = -2;     // LDS bytes of a call-kernel workgroup without coefficient area (fit_budget; -2: not asked yet)
    bool wide = false;         // more than eight samples, four l2fc terms or four nested ranges on a path: the wide build of the kernels
    vlr::DevPlanT<8> host8{};  // the same plan in the layout of the standard and deep builds (plans of at most eight samples)
    vlr::DevPlanT<16>* dev = nullptr;
    void* blob = nullptr;      // device arrays
    int max_depth_per_sample = 200;  // reference default max_depth (src/variants/sample.rs:236)
    int max_obs = 0;                 // 0: max_depth_per_sample * S
    int n_events = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    // staging for vlr_batch_run_host: two slots (buffer + stream) so that the copies of one chunk of loci overlap the
    // kernel of the previous one
    void* stage[2] = {nullptr, nullptr};
    size_t stage_bytes[2] = {0, 0};
    hipStream_t stage_stream[2] = {nullptr, nullptr};
    hipStream_t afd_aux_stream = nullptr;   // second lane of the AFD sub-ranges (vlr_batch_run)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned long long* work_dev = nullptr;
    // AFD replay scratch (device), one per slot: is_discrete mask of the MAP, and marginal/best_event when the caller passes NULL
    void* afd_scratch[2] = {nullptr, nullptr};
    size_t afd_scratch_bytes[2] = {0, 0};
    // kernel scratch, one per slot: the third likelihood coefficient of every kept observation (8 B x max_obs per locus)
    void* escratch[2] = {nullptr, nullptr};
    size_t escratch_bytes[2] = {0, 0};
    // AFD log, one per slot (only when AFD lists are requested): afd_log_words 8-byte words per locus
    void* afd_log[2] = {nullptr, nullptr};
    size_t afd_log_bytes[2] = {0, 0};
    size_t afd_log_words = 0;
    void* deep_pool[2] = {nullptr, nullptr};  // coefficient triples of the loci above the LDS budget (deep launch), one pool per slot;
    size_t deep_pool_bytes[2] = {0, 0};       // its first 128 bytes hold the bump counters of the two AFD lanes
    void* afd_keys[2] = {nullptr, nullptr};   // l2fc-list keys of the AFD entries (DevResults::afd_key), one per slot
    size_t afd_keys_bytes[2] = {0, 0};
    int64_t afd_log_loci[2] = {0, 0};  // loci whose log regions fit afd_log[slot] (the budget of ensure_buffers)
    int slot = 0;  // slot used by the next vlr_batch_run (set by vlr_batch_run_host)
    // does the next vlr_batch_run need the deep launch (a pileup above the LDS budget)?  -1: unknown (device pointers: the host
    // cannot see the depths, pool and launch as a precaution), 0 / 1: vlr_batch_run_host has looked at the chunk's offsets — the
    // 512 MiB pool is only allocated, and the second launch only enqueued, when a locus needs them
    int deep_hint = -1;
};

namespace {

HostSpectrum host_spectrum(const vlr_spectrum& s, const double* pool) {
    HostSpectrum h;
    h.is_set = s.kind == VLR_SPECTRUM_SET;
    if (h.is_set) h.set.assign(pool + s.set_offset, pool + s.set_offset + s.set_len);
    h.start = s.start; h.end = s.end; h.lex = s.left_exclusive != 0; h.rex = s.right_exclusive != 0;
    return h;
}

// Tabulate Prior::compute over per-sample VAF classes:
//   uniform sample      : {in universe & 0, in universe & != 0, outside}
//   ploidy-based sample : {k/ploidy for k = 0..ploidy, any other value}
int build_prior_table(const vlr_scenario_desc* d, vlr::DevPlanT<16>& P, std::vector<double>& table) {
    const int S = d->n_samples;
    HostPrior pr;
    pr.S = S;
    pr.uniform.assign(d->uniform_prior, d->uniform_prior + S);
    pr.ploidy.assign(d->ploidy, d->ploidy + S);
    pr.germline_rate.assign(d->germline_mutation_rate, d->germline_mutation_rate + S);
    pr.somatic_rate.assign(d->somatic_effective_mutation_rate, d->somatic_effective_mutation_rate + S);
    pr.inh.assign(d->inheritance, d->inheritance + S);
    pr.het_ln = std::isnan(d->heterozygosity) ? NAN : std::log(d->heterozygosity);
    pr.var_het_ln = d->variant_heterozygosity_ln;
    pr.var_som_ln = d->variant_somatic_effective_mutation_rate_ln;
    pr.f_indel = d->fraction_indel; pr.f_mnv = d->fraction_mnv; pr.f_sv = d->fraction_sv;
    pr.absent_only = d->is_absent_only != 0;
    pr.universe.resize(S);
    for (int s = 0; s < S; ++s)
        for (int i = d->universe_offset[s]; i < d->universe_offset[s + 1]; ++i) pr.universe[s].push_back(host_spectrum(d->universe[i], d->vafs));

    // CheckablePrior::check (prior.rs:788-825)
    for (int s = 0; s < S; ++s) {
        const vlr_inheritance& in = d->inheritance[s];
        if (in.kind == VLR_INHERIT_NONE) continue;
        auto has_ploidy = [&](int x) { return x >= 0 && x < S && d->ploidy[x] >= 0; };
        if (in.kind == VLR_INHERIT_MENDELIAN && (!has_ploidy(in.from0) || !has_ploidy(in.from1)))
            return fail(VLR_ERR_INVALID_PRIOR, "inheritance defined but parental samples do not have a ploidy");
        if (in.kind != VLR_INHERIT_MENDELIAN && !has_ploidy(in.from0))
            return fail(VLR_ERR_INVALID_PRIOR, "inheritance defined but parental samples do not have a ploidy");
        if (in.kind == VLR_INHERIT_MENDELIAN && std::isnan(d->germline_mutation_rate[s]))
            return fail(VLR_ERR_INVALID_PRIOR, "mendelian inheritance but no germline mutation rate defined");
        if (in.kind == VLR_INHERIT_SUBCLONAL && std::isnan(d->somatic_effective_mutation_rate[s]))
            return fail(VLR_ERR_INVALID_PRIOR, "subclonal inheritance defined but no somatic mutation");
        // Clonal inheritance of the somatic VAF by a sample WITHOUT own somatic rate (prior.rs:489-499) compares the
        // effective somatic VAFs of sample and parent.  The sample has no somatic variation, so the recursion of calc_prob
        // (prior.rs:417-427) pins its germline VAF to its VAF: its effective somatic VAF is exactly zero and the comparison
        // reduces to "the parent's VAF is one of its germline levels" — a property of the parent's class.  The tabulated
        // prior is therefore exact for this case too (round 1 rejected it with VLR_ERR_UNSUPPORTED).
    }

    std::vector<std::vector<double>> reps(S);  // representative VAF per class; NaN = class is impossible (-inf)
    size_t size = 1;
    for (int s = 0; s < S; ++s) {
        if (d->uniform_prior[s]) {
            P.prior_kind[s] = vlr::PK_UNIFORM;
            double nonzero = NAN;
            for (auto& sp : pr.universe[s]) {
                if (!std::isnan(nonzero)) break;
                if (sp.is_set) {
                    for (double v : sp.set)
                        if (v != 0.0) { nonzero = v; break; }
                } else {
                    for (double v : {(sp.start + sp.end) / 2.0, sp.end, sp.start})
                        if (v != 0.0 && sp.contains(v)) { nonzero = v; break; }
                }
            }
            bool zero_in = false;
            for (auto& sp : pr.universe[s]) zero_in = zero_in || sp.contains(0.0);
            reps[s] = {zero_in ? 0.0 : NAN, nonzero, NAN};
        } else {
            int pl = d->ploidy[s];
            bool somatic = !std::isnan(d->somatic_effective_mutation_rate[s]);
            if (pl < 0) return fail(VLR_ERR_INVALID_PRIOR, "sample %d has neither universe nor ploidy (grammar/mod.rs:569-574)", s);
            if (!somatic && std::isnan(d->heterozygosity))
                return fail(VLR_ERR_INVALID_PRIOR, "bug: not enough info for prior but no universe specified (prior.rs:433)");
            P.prior_kind[s] = somatic ? vlr::PK_SOMATIC : vlr::PK_GERMLINE;
            for (int k = 0; k <= pl; ++k) reps[s].push_back(pl > 0 ? (double)k / pl : 0.0);
            reps[s].push_back(pl > 0 ? 0.37 / pl : 0.37);  // "other": between 0 and 1/ploidy, never k/ploidy
        }
        P.ploidy[s] = d->ploidy[s];
        P.n_class[s] = (int)reps[s].size();
        P.class_stride[s] = (int)size;
        size *= reps[s].size();
        if (size > (1u << 20)) return fail(VLR_ERR_UNSUPPORTED, "prior table too large");
    }
    P.table_size = (int)size;
    table.assign((size_t)vlr::kNVariantTypes * size, NEG_INF);
    for (int vt = 0; vt < vlr::kNVariantTypes; ++vt) {
        pr.vt = vt;
        for (size_t idx = 0; idx < size; ++idx) {
            std::vector<double> ev(S);
            bool possible = true;
            size_t rem = idx;
            for (int s = 0; s < S; ++s) {
                int cls = (int)(rem % reps[s].size());
                rem /= reps[s].size();
                ev[s] = reps[s][cls];
                if (std::isnan(ev[s])) possible = false;
            }
            if (!possible) continue;
            bool ok = true;
            double v = pr.compute(ev, &ok);
            if (!ok) return fail(VLR_ERR_INVALID_PRIOR, "prior cannot be evaluated (ploidies of child and parents do not match, or missing rates)");
            table[(size_t)vt * size + idx] = v;
        }
    }
    return VLR_OK;
}

int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VLR_ERR_NO_DEVICE, "no HIP device available (%s); the engine has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(VLR_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, n);
    return VLR_OK;
}

}  // namespace

extern "C" {

int vlr_abi_version(void) { return VLR_ABI_VERSION; }
#ifndef VLR_SRC_ID
#define VLR_SRC_ID "unknown"
#endif
const char* vlr_build_id(void) { return VLR_SRC_ID; }
const char* vlr_last_error(void) { return g_err.c_str(); }
void vlr_set_error(const char* msg) { g_err = msg ? msg : ""; }  // for the other translation units of the library (vlr_ingest.cpp)

int vlr_plan_create(const vlr_scenario_desc* d, int device, vlr_plan** out) {
    using namespace vlr;
    if (!d || !out) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const int S = d->n_samples;
    if (S < 1 || S > kMaxSamples) return fail(VLR_ERR_INVALID_ARGUMENT, "n_samples %d outside 1..%d", S, kMaxSamples);
    // (more than kMaxNamedEvents = 30 events: the wide build of the kernels, whose masks of event groups are 64 bits)
    if (d->n_events < 0 || d->n_events > kMaxNamedEventsWide) return fail(VLR_ERR_UNSUPPORTED, "n_events %d outside 0..%d", d->n_events, kMaxNamedEventsWide);
    for (int s = 0; s < S; ++s) {
        if (!(d->resolution[s] > 0.0 && d->resolution[s] < 1.0)) return fail(VLR_ERR_INVALID_ARGUMENT, "resolution must be in (0,1) (grammar/mod.rs:477-481)");
        int by = d->contaminated_by[s];
        if (by >= S || by == s) return fail(VLR_ERR_INVALID_ARGUMENT, "invalid contamination sample (errors::Error::InvalidContaminationSampleName)");
        if (by >= 0) {
            double purity = 1.0 - d->contamination_fraction[s];
            if (!(purity > 0.0 && purity <= 1.0)) return fail(VLR_ERR_INVALID_ARGUMENT, "purity must be in (0,1] (likelihood.rs:78)");
        }
    }
    DevPlanT<16> P{};
    P.S = S;
    P.n_named = d->n_events;
    P.n_univ = 1 + 2 * d->n_events;
    for (int s = 0; s < kMaxSamples; ++s) { P.by[s] = -1; P.rho[s] = 1.0; P.irho[s] = 0.0; P.resolution[s] = 0.01; }
    for (int s = 0; s < S; ++s) {
        P.resolution[s] = d->resolution[s];
        P.by[s] = d->contaminated_by[s];
        if (P.by[s] >= 0) { P.rho[s] = 1.0 - d->contamination_fraction[s]; P.irho[s] = 1.0 - P.rho[s]; }
        P.uni_off[s] = d->universe_offset[s];
    }
    P.uni_off[S] = d->universe_offset[S];

    // ---- nodes (+ VAFTree::absent chain, grammar/vaftree.rs:18-40)
    std::vector<DevNode> nodes(d->n_nodes + S);
    std::vector<int32_t> child;
    auto conv_spec = [&](const vlr_spectrum& s) {
        DevSpectrum o{};
        o.kind = s.kind; o.set_off = s.set_offset; o.set_len = s.set_len;
        o.lex = s.left_exclusive; o.rex = s.right_exclusive; o.start = s.start; o.end = s.end;
        return o;
    };
    for (int i = 0; i < d->n_nodes; ++i) {
        const vlr_node& n = d->nodes[i];
        DevNode& o = nodes[i];
        o = DevNode{};
        o.kind = n.kind; o.sample = n.sample; o.sample_b = n.sample_b; o.cmp = n.cmp; o.lfc_value = n.lfc_value;
        o.vafs = conv_spec(n.vafs);
        o.positive = n.positive; o.refbase = n.refbase; o.altbase = n.altbase;
        o.child_off = (int)child.size(); o.n_children = n.n_children;
        for (int c = 0; c < n.n_children; ++c) {
            int ci = d->child_index[n.child_offset + c];
            if (ci < 0 || ci >= d->n_nodes) return fail(VLR_ERR_INVALID_ARGUMENT, "child index out of range");
            child.push_back(ci);
        }
        if (n.kind == VLR_NODE_SAMPLE) {
            if (n.sample < 0 || n.sample >= S) return fail(VLR_ERR_INVALID_ARGUMENT, "invalid sample index in VAF tree (errors::Error::InvalidSampleName)");
            if (n.vafs.kind == VLR_SPECTRUM_SET && n.vafs.set_len > kMaxSet) return fail(VLR_ERR_UNSUPPORTED, "VAF set larger than %d", kMaxSet);
        }
        if (n.kind == VLR_NODE_LFC && (n.sample < 0 || n.sample >= S || n.sample_b < 0 || n.sample_b >= S))
            return fail(VLR_ERR_INVALID_ARGUMENT, "invalid sample index in l2fc term");
    }
    // vafs pool: copy + one 0.0 for the absent chain
    int pool_len = 0;
    for (int i = 0; i < d->n_nodes; ++i)
        if (d->nodes[i].kind == VLR_NODE_SAMPLE && d->nodes[i].vafs.kind == VLR_SPECTRUM_SET)
            pool_len = std::max(pool_len, d->nodes[i].vafs.set_offset + d->nodes[i].vafs.set_len);
    for (int i = 0; i < d->universe_offset[S]; ++i)
        if (d->universe[i].kind == VLR_SPECTRUM_SET) pool_len = std::max(pool_len, d->universe[i].set_offset + d->universe[i].set_len);
    std::vector<double> pool(d->vafs, d->vafs + pool_len);
    pool.push_back(0.0);
    for (int s = 0; s < S; ++s) {
        DevNode& o = nodes[d->n_nodes + s];
        o = DevNode{};
        o.kind = VLR_NODE_SAMPLE; o.sample = s;
        o.vafs.kind = VLR_SPECTRUM_SET; o.vafs.set_off = pool_len; o.vafs.set_len = 1;
        o.child_off = (int)child.size();
        o.n_children = (s + 1 < S) ? 1 : 0;
        if (s + 1 < S) child.push_back(d->n_nodes + s + 1);
    }
    P.absent_root = d->n_nodes;
    P.n_nodes = (int)nodes.size();
    std::vector<int32_t> roots(d->root_index, d->root_index + d->event_root_offset[d->n_events]);
    std::vector<int32_t> root_off(d->event_root_offset, d->event_root_offset + d->n_events + 1);
    for (int r : roots)
        if (r < 0 || r >= d->n_nodes) return fail(VLR_ERR_INVALID_ARGUMENT, "root index out of range");

    // ---- static limits of the device walk: range nesting, LFC terms and frames per path
    int max_range = 0, max_lfc = 0, max_frames = 0, max_tab = 0;
    {
        struct It { int node, ranges, lfcs, frames; };
        std::vector<It> st;
        for (int r : roots) st.push_back({r, 0, 0, 0});
        size_t guard = 0;
        while (!st.empty()) {
            It it = st.back();
            st.pop_back();
            if (++guard > 10000000) return fail(VLR_ERR_INVALID_ARGUMENT, "VAF tree too large or cyclic");
            const DevNode& n = nodes[it.node];
            if (n.kind == VLR_NODE_SAMPLE) {
                it.frames++;
                if (n.vafs.kind == VLR_SPECTRUM_RANGE && !(n.vafs.start == n.vafs.end)) {
                    if (n.n_children > 0) max_tab = std::max(max_tab, it.ranges + 1);
                    it.ranges++;
                }
            }
            if (n.kind == VLR_NODE_LFC) it.lfcs++;
            if (n.n_children > 1) it.frames++;
            max_range = std::max(max_range, it.ranges);
            max_lfc = std::max(max_lfc, it.lfcs);
            max_frames = std::max(max_frames, it.frames);
            for (int c = 0; c < n.n_children; ++c) st.push_back({child[n.child_off + c], it.ranges, it.lfcs, it.frames});
        }
    }
    if (max_range > kMaxRangeDepthWide) return fail(VLR_ERR_UNSUPPORTED, "more than %d nested VAF ranges on one path", kMaxRangeDepthWide);
    if (max_lfc > kMaxLfcWide) return fail(VLR_ERR_UNSUPPORTED, "more than %d l2fc terms on one path", kMaxLfcWide);
    // plans beyond the standard build's limits run the wide build of the kernels
    const bool needs_wide = S > 8 || max_range > kMaxRangeDepthStd || max_lfc > kMaxLfcStd || d->n_events > kMaxNamedEvents;
    if (max_frames > kMaxFrames) return fail(VLR_ERR_UNSUPPORTED, "VAF tree deeper than %d frames", kMaxFrames);
    P.max_range_depth = std::max(1, max_range);
    P.max_tab_depth = max_tab;
    P.max_set = 1;
    for (const DevNode& n : nodes)
        if (n.kind == VLR_NODE_SAMPLE && n.vafs.kind == VLR_SPECTRUM_SET) P.max_set = std::max(P.max_set, n.vafs.set_len);
    P.max_frames = std::max(S, max_frames);  // the absent chain pushes one frame per sample
    {
        // capacity of a visited-point table: 2 endpoints + 3 per bisection round + 7 tail points, with at most
        // ceil(ln(1/res) / ln(4/3)) + 1 rounds (the bracket shrinks to at most 3/4 per round,
        // utils/adaptive_integration.rs:61-94); at least the 11 Simpson points
        int cap = 16;
        for (int s = 0; s < S; ++s) {
            int rounds = (int)std::ceil(std::log(1.0 / d->resolution[s]) / std::log(4.0 / 3.0)) + 1;
            cap = std::max(cap, 2 + 3 * rounds + 7);
        }
        cap = std::min((cap + 3) & ~3, (int)kTableCapMax);
        P.table_cap = cap;
    }

    std::vector<DevSpectrum> uni;
    for (int i = 0; i < d->universe_offset[S]; ++i) uni.push_back(conv_spec(d->universe[i]));
    // per (group, sample): distinct spectra of the Sample nodes in the group's trees
    std::vector<int32_t> gs_off(1, 0);
    std::vector<DevSpectrum> gs;
    for (int g = 0; g <= d->n_events; ++g) {
        std::vector<int> reach;
        std::vector<int> stack;
        if (g == 0) stack.push_back(P.absent_root);
        else for (int ri = root_off[g - 1]; ri < root_off[g]; ++ri) stack.push_back(roots[ri]);
        std::vector<char> seen(nodes.size(), 0);
        while (!stack.empty()) {
            int n = stack.back();
            stack.pop_back();
            if (seen[n]) continue;
            seen[n] = 1;
            reach.push_back(n);
            for (int c = 0; c < nodes[n].n_children; ++c) stack.push_back(child[nodes[n].child_off + c]);
        }
        for (int s = 0; s < S; ++s) {
            size_t first = gs.size();
            for (int n : reach) {
                if (nodes[n].kind != VLR_NODE_SAMPLE || nodes[n].sample != s) continue;
                const DevSpectrum& sp = nodes[n].vafs;
                bool dup = false;
                for (size_t k = first; k < gs.size(); ++k)
                    if (memcmp(&gs[k], &sp, sizeof(DevSpectrum)) == 0) dup = true;
                if (!dup) gs.push_back(sp);
            }
            gs_off.push_back((int32_t)gs.size());
        }
    }
    // per Sample node: the groups whose spectra for its sample overlap the node's own spectrum (DevNode::alive_mask)
    {
        auto overlap = [&](const DevSpectrum& a, const DevSpectrum& b) {
            const double eps = 1e-9;
            auto lo = [&](const DevSpectrum& x, int i) { return x.kind == VLR_SPECTRUM_SET ? pool[x.set_off + i] : x.start; };
            auto hi = [&](const DevSpectrum& x, int i) { return x.kind == VLR_SPECTRUM_SET ? pool[x.set_off + i] : x.end; };
            const int na = a.kind == VLR_SPECTRUM_SET ? a.set_len : 1, nb = b.kind == VLR_SPECTRUM_SET ? b.set_len : 1;
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < nb; ++j)
                    if (lo(a, i) <= hi(b, j) + eps && lo(b, j) <= hi(a, i) + eps) return true;
            return false;
        };
        for (DevNode& n : nodes) {
            n.alive_mask = 0; n.alive_mask_hi = 0;
            if (n.kind != VLR_NODE_SAMPLE) continue;
            uint64_t am = 0;
            for (int g = 0; g <= d->n_events; ++g)
                for (int k = gs_off[g * S + n.sample]; k < gs_off[g * S + n.sample + 1]; ++k)
                    if (overlap(n.vafs, gs[k])) { am |= 1ull << g; break; }
            n.alive_mask = (int32_t)(uint32_t)am; n.alive_mask_hi = (int32_t)(uint32_t)(am >> 32);
        }
    }
    std::vector<double> table;
    int rc = build_prior_table(d, P, table);
    if (rc != VLR_OK) return rc;

    // ---- all-discrete roots, flattened for the lane-parallel leaf evaluation (vlr_kernels.hip eval_discrete_root)
    std::vector<DevDLeaf> dleaf;
    std::vector<DevDKey> dkey;
    // ---- compiled roots (DevFastRoot): chains of single-valued Sample nodes ending in a leaf Range node
    std::vector<DevFastRoot> froot(1 + roots.size());
    for (auto& f : froot) { memset(&f, 0, sizeof f); }
    std::vector<int32_t> droot(2 * (1 + roots.size()), -1);
    {
        auto node_values = [&](const DevNode& n, std::vector<double>& vals) {
            vals.clear();
            if (n.kind != VLR_NODE_SAMPLE) return false;
            if (n.vafs.kind == VLR_SPECTRUM_SET) vals.assign(pool.begin() + n.vafs.set_off, pool.begin() + n.vafs.set_off + n.vafs.set_len);
            else if (n.vafs.start == n.vafs.end && !n.vafs.lex && !n.vafs.rex) vals.push_back(n.vafs.start);
            return !vals.empty();
        };
        auto spec_contains = [&](const DevSpectrum& sp, double v) {
            if (sp.kind == VLR_SPECTRUM_SET) {
                for (int i = 0; i < sp.set_len; ++i) if (pool[sp.set_off + i] == v) return true;
                return false;
            }
            bool lo = sp.lex ? sp.start < v : sp.start <= v, hi = sp.rex ? sp.end > v : sp.end >= v;
            return lo && hi;
        };
        auto rel_eq = [](double a, double b) {
            const double eps = 2.220446049250313e-16;
            if (a == b) return true;
            if (std::isinf(a) || std::isinf(b)) return false;
            double df = std::fabs(a - b);
            if (df <= eps) return true;
            return df <= std::max(std::fabs(a), std::fabs(b)) * eps;
        };
        auto prior_class = [&](int s2, double v) {  // same classes as the device's prior_class()
            if (P.prior_kind[s2] == PK_UNIFORM) {
                bool in = false;
                for (int u = P.uni_off[s2]; u < P.uni_off[s2 + 1]; ++u) in = in || spec_contains(uni[u], v);
                return in ? (v == 0.0 ? 0 : 1) : 2;
            }
            int pl = P.ploidy[s2];
            double dp = (double)pl, k = std::rint(dp * v);
            bool match;
            if (P.prior_kind[s2] == PK_GERMLINE) match = rel_eq(dp * v, k);
            else match = pl > 0 ? rel_eq(v - k / dp, 0.0) : (v == 0.0);
            return (match && k >= 0.0 && k <= dp) ? (int)k : pl + 1;
        };
        // VAFTree::contains (vaftree.rs:42-51,116-164) for operands without l2fc terms
        std::function<bool(int, const double*)> contains = [&](int node, const double* v) -> bool {
            const DevNode& n = nodes[node];
            bool in;
            if (n.kind == VLR_NODE_SAMPLE) in = spec_contains(n.vafs, v[n.sample]);
            else if (n.kind == VLR_NODE_LFC) in = false;
            else in = n.kind != VLR_NODE_FALSE;
            if (!in) return false;
            if (n.n_children == 0) return true;
            for (int ci = 0; ci < n.n_children; ++ci)
                if (contains(child[n.child_off + ci], v)) return true;
            return false;
        };
        auto group_contains = [&](int g, const double* v) {
            if (g == 0) return contains(P.absent_root, v);
            for (int ri = root_off[g - 1]; ri < root_off[g]; ++ri)
                if (contains(roots[ri], v)) return true;
            return false;
        };
        struct Path { double vaf[kMaxSamples]; uint32_t have, posmask; };
        const uint32_t full = (1u << S) - 1u;
        std::vector<Path> paths;
        std::function<bool(int, Path)> flatten = [&](int node, Path cur) -> bool {
            const DevNode& n = nodes[node];
            std::vector<double> vals;
            if (!node_values(n, vals)) return false;
            const int s2 = n.sample;
            if (cur.have & (1u << s2)) return false;
            bool allpos = true;
            for (double v : vals) allpos = allpos && v > 0.0;
            for (double v : vals) {
                Path q = cur;
                q.vaf[s2] = v; q.have |= 1u << s2;
                if (allpos) q.posmask |= 1u << s2;
                if (n.n_children == 0) {
                    if (q.have != full || paths.size() >= (size_t)kMaxDLeaf) return false;
                    paths.push_back(q);
                } else {
                    for (int ci = 0; ci < n.n_children; ++ci)
                        if (!flatten(child[n.child_off + ci], q)) return false;
                }
            }
            return true;
        };
        bool keys_ok = true;
        for (size_t i = 0; i < 1 + roots.size() && keys_ok; ++i) {
            const int root = i == 0 ? P.absent_root : roots[i - 1];
            int own = 0;
            if (i > 0) for (int e = 0; e < d->n_events; ++e) if ((int)(i - 1) >= root_off[e] && (int)(i - 1) < root_off[e + 1]) own = e + 1;
            paths.clear();
            Path start{};
            if (!flatten(root, start) || paths.empty()) continue;
            const size_t first = dleaf.size();
            for (const Path& q : paths) {
                DevDLeaf L{};
                int idx = 0;
                for (int s2 = 0; s2 < S; ++s2) {
                    L.vaf[s2] = q.vaf[s2];
                    idx += prior_class(s2, q.vaf[s2]) * P.class_stride[s2];
                    const double a = q.vaf[s2], bq = P.by[s2] >= 0 ? q.vaf[P.by[s2]] : 0.0;
                    int k = -1;
                    for (size_t j = 0; j < dkey.size(); ++j)
                        if (dkey[j].sample == s2 && dkey[j].a == a && dkey[j].b == bq) k = (int)j;
                    if (k < 0) {
                        if (dkey.size() >= (size_t)kMaxDKeys) { keys_ok = false; break; }
                        DevDKey nk{};
                        nk.sample = s2; nk.a = a; nk.b = bq;
                        k = (int)dkey.size();
                        dkey.push_back(nk);
                    }
                    L.key[s2] = (uint8_t)k;
                }
                if (!keys_ok) break;
                L.prior_idx = idx;
                L.posmask = q.posmask;
                uint64_t cm = 0;
                for (int g = 0; g <= d->n_events; ++g)
                    if (g != own && group_contains(g, q.vaf)) cm |= 1ull << g;
                L.cmask = (uint32_t)cm; L.cmask_hi = (uint32_t)(cm >> 32);
                dleaf.push_back(L);
            }
            if (!keys_ok) break;
            droot[2 * i] = (int32_t)first;
            droot[2 * i + 1] = (int32_t)dleaf.size();
        }
        if (!keys_ok || getenv("VLR_NO_DISCRETE_ROOTS")) {  // too many distinct likelihoods (or switched off for A/B runs): general walk
            dleaf.clear(); dkey.clear();
            std::fill(droot.begin(), droot.end(), -1);
        }
        P.n_dkey = (int32_t)dkey.size();
        P.n_dleaf = (int32_t)dleaf.size();
        if (!getenv("VLR_NO_FAST_ROOTS")) {
            auto may_contain = [&](int g, int s2, double v) {  // group_may_contain of the kernel: some spectrum of group g for sample s2 holds v
                for (int k = gs_off[g * S + s2]; k < gs_off[g * S + s2 + 1]; ++k) {
                    const DevSpectrum& sp = gs[k];
                    if (sp.kind == VLR_SPECTRUM_SET) { for (int i = 0; i < sp.set_len; ++i) if (pool[sp.set_off + i] == v) return true; }
                    else if ((sp.start < v || (!sp.lex && sp.start == v)) && (sp.end > v || (!sp.rex && sp.end == v))) return true;
                }
                return false;
            };
            for (size_t i = 0; i < froot.size(); ++i) {
                int own = 0;
                if (i > 0) for (int e = 0; e < d->n_events; ++e) if ((int)(i - 1) >= root_off[e] && (int)(i - 1) < root_off[e + 1]) own = e + 1;
                DevFastRoot f{};
                int node = i == 0 ? P.absent_root : roots[i - 1];
                uint32_t have = 0;
                uint64_t alive = ((1ull << (d->n_events + 1)) - 1ull) & ~(1ull << own);
                auto node_alive = [](const DevNode& n) { return (uint64_t)(uint32_t)n.alive_mask | ((uint64_t)(uint32_t)n.alive_mask_hi << 32); };
                int pidx = 0;
                bool ok = true;
                for (;;) {
                    const DevNode& n = nodes[node];
                    if (n.kind != VLR_NODE_SAMPLE || (have & (1u << n.sample))) { ok = false; break; }
                    have |= 1u << n.sample;
                    const bool single = n.vafs.kind == VLR_SPECTRUM_SET ? n.vafs.set_len == 1 : (n.vafs.start == n.vafs.end && !n.vafs.lex && !n.vafs.rex);
                    if (n.n_children == 0) {   // the leaf: a proper range
                        if (n.vafs.kind != VLR_SPECTRUM_RANGE || n.vafs.start == n.vafs.end) { ok = false; break; }
                        f.inner = n.sample; f.leaf_node = node;
                        f.start = n.vafs.start; f.end = n.vafs.end; f.lex = n.vafs.lex; f.rex = n.vafs.rex;
                        f.alive = (int32_t)(uint32_t)(alive & node_alive(n)); f.alive_hi = (int32_t)(uint32_t)((alive & node_alive(n)) >> 32);
                        break;
                    }
                    if (n.n_children != 1 || !single || f.n_fixed >= kMaxSamples) { ok = false; break; }
                    const double v = n.vafs.kind == VLR_SPECTRUM_SET ? pool[n.vafs.set_off] : n.vafs.start;
                    f.fsample[f.n_fixed] = n.sample; f.fvaf[f.n_fixed] = v; f.n_fixed++;
                    f.disc |= 1 << n.sample;
                    pidx += prior_class(n.sample, v) * P.class_stride[n.sample];
                    // walk_root: f.sv_alive = c.alive & nd.alive_mask; c.alive = alive_update(sv_alive & nd.alive_mask, s, v)
                    uint64_t m = alive & node_alive(n), res = m;
                    for (int g = 0; g <= d->n_events; ++g)
                        if (((m >> g) & 1) && !may_contain(g, n.sample, v)) res &= ~(1ull << g);
                    alive = res;
                    node = child[n.child_off];
                }
                if (!ok || have != ((1u << S) - 1u)) {   // (a chain record needs every sample on the path: the operands of a leaf are complete)
                    // Not a chain.  If the probe pass would hand the root to the general pass at its very first node anyway — a Sample
                    // node with a proper Range above other nodes, or with a Set of several members (walk_root: deferred = 2) — say so:
                    // the event loop then skips the probe walk (kind 2).
                    const DevNode& r0 = nodes[i == 0 ? P.absent_root : roots[i - 1]];
                    if (r0.kind == VLR_NODE_SAMPLE &&
                        ((r0.vafs.kind == VLR_SPECTRUM_RANGE && r0.vafs.start != r0.vafs.end && r0.n_children != 0) ||
                         (r0.vafs.kind == VLR_SPECTRUM_SET && r0.vafs.set_len > 1)))
                        froot[i].kind = 2;
                    continue;
                }
                f.kind = 1; f.pidx = pidx;
                froot[i] = f;
            }
        }
    }

    rc = check_device(device);
    if (rc != VLR_OK) return rc;
    HIP_TRY(hipSetDevice(device));

    // ---- upload
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_nodes = 0, o_child = o_nodes + al(nodes.size() * sizeof(DevNode)), o_pool = o_child + al(std::max<size_t>(1, child.size()) * 4),
           o_roots = o_pool + al(pool.size() * 8), o_roff = o_roots + al(std::max<size_t>(1, roots.size()) * 4),
           o_uni = o_roff + al(root_off.size() * 4), o_tab = o_uni + al(std::max<size_t>(1, uni.size()) * sizeof(DevSpectrum)),
           o_gso = o_tab + al(table.size() * 8), o_gs = o_gso + al(gs_off.size() * 4),
           o_dl = o_gs + al(std::max<size_t>(1, gs.size()) * sizeof(DevSpectrum)),
           o_dk = o_dl + al(std::max<size_t>(1, dleaf.size()) * sizeof(DevDLeaf)),
           o_dr = o_dk + al(std::max<size_t>(1, dkey.size()) * sizeof(DevDKey)),
           o_fr = o_dr + al(droot.size() * 4),
           total = o_fr + al(froot.size() * sizeof(DevFastRoot));
    std::vector<char> hostblob(total, 0);
    memcpy(&hostblob[o_nodes], nodes.data(), nodes.size() * sizeof(DevNode));
    if (!child.empty()) memcpy(&hostblob[o_child], child.data(), child.size() * 4);
    memcpy(&hostblob[o_pool], pool.data(), pool.size() * 8);
    if (!roots.empty()) memcpy(&hostblob[o_roots], roots.data(), roots.size() * 4);
    memcpy(&hostblob[o_roff], root_off.data(), root_off.size() * 4);
    if (!uni.empty()) memcpy(&hostblob[o_uni], uni.data(), uni.size() * sizeof(DevSpectrum));
    memcpy(&hostblob[o_tab], table.data(), table.size() * 8);
    memcpy(&hostblob[o_gso], gs_off.data(), gs_off.size() * 4);
    if (!gs.empty()) memcpy(&hostblob[o_gs], gs.data(), gs.size() * sizeof(DevSpectrum));
    if (!dleaf.empty()) memcpy(&hostblob[o_dl], dleaf.data(), dleaf.size() * sizeof(DevDLeaf));
    if (!dkey.empty()) memcpy(&hostblob[o_dk], dkey.data(), dkey.size() * sizeof(DevDKey));
    memcpy(&hostblob[o_dr], droot.data(), droot.size() * 4);
    memcpy(&hostblob[o_fr], froot.data(), froot.size() * sizeof(DevFastRoot));

    vlr_plan* plan = new vlr_plan();
    plan->device = device;
    plan->n_events = d->n_events;
    if (hipMalloc(&plan->blob, total) != hipSuccess) { delete plan; return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", total); }
    char* base = (char*)plan->blob;
    P.nodes = (const DevNode*)(base + o_nodes);
    P.child_index = (const int32_t*)(base + o_child);
    P.vafs = (const double*)(base + o_pool);
    P.roots = (const int32_t*)(base + o_roots);
    P.root_off = (const int32_t*)(base + o_roff);
    P.universe = (const DevSpectrum*)(base + o_uni);
    P.prior_table = (const double*)(base + o_tab);
    P.grp_spec_off = (const int32_t*)(base + o_gso);
    P.grp_spec = (const DevSpectrum*)(base + o_gs);
    P.dleaf = (const DevDLeaf*)(base + o_dl);
    P.dkey = (const DevDKey*)(base + o_dk);
    P.droot = (const int32_t*)(base + o_dr);
    P.froot = (const DevFastRoot*)(base + o_fr);
    plan->host = P;
    plan->wide = needs_wide;
    if (S <= 8) plan->host8 = narrow_plan(P);
    {   // The tables of the plan (Set candidates, the replay's lists of seen discrete operands: S x max_set doubles each) must leave room
        // for at least a minimal pileup in the 160 KiB of LDS a CU has; otherwise the first batch would fail with a launch error.
        const long long floor_b = !needs_wide ? vlr_plan_lds_floor(&plan->host8, P.n_univ, S, P.max_range_depth) : vlr_plan_lds_floor_wide(&plan->host, P.n_univ, S, P.max_range_depth);
        const long long min_pileup = 16ll * 64;   // coefficient pairs of 64 observations
        if (floor_b + min_pileup > 160ll * 1024) {
            vlr_plan_destroy(plan);
            return fail(VLR_ERR_UNSUPPORTED, "scenario needs %lld bytes of LDS per locus before any observation (largest Set spectrum %d members x %d samples): "
                                             "above the 160 KiB of a CU", floor_b, P.max_set, S);
        }
    }
    hipError_t e = hipMemcpy(plan->blob, hostblob.data(), total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&plan->dev, sizeof(DevPlanT<16>));
    if (e == hipSuccess) e = hipMemcpy(plan->dev, &P, sizeof(DevPlanT<16>), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&plan->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&plan->ev_stop);
    if (e == hipSuccess) e = hipMalloc((void**)&plan->work_dev, 64 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(plan->work_dev, 0, 64 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        vlr_plan_destroy(plan);
        return fail(VLR_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
    }
    *out = plan;
    return VLR_OK;
}

void vlr_plan_destroy(vlr_plan* plan) {
    if (!plan) return;
    if (plan->blob) (void)hipFree(plan->blob);
    if (plan->dev) (void)hipFree(plan->dev);
    for (int k = 0; k < 2; ++k) {
        if (plan->stage[k]) (void)hipFree(plan->stage[k]);
        if (plan->afd_scratch[k]) (void)hipFree(plan->afd_scratch[k]);
        if (plan->escratch[k]) (void)hipFree(plan->escratch[k]);
        if (plan->afd_log[k]) (void)hipFree(plan->afd_log[k]);
        if (plan->afd_keys[k]) (void)hipFree(plan->afd_keys[k]);
        if (plan->deep_pool[k]) (void)hipFree(plan->deep_pool[k]);
        if (plan->stage_stream[k]) (void)hipStreamDestroy(plan->stage_stream[k]);
    }
    if (plan->work_dev) (void)hipFree(plan->work_dev);
    if (plan->afd_aux_stream) (void)hipStreamDestroy(plan->afd_aux_stream);
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    if (plan->ev_join) (void)hipEventDestroy(plan->ev_join);
    if (plan->ev_start) (void)hipEventDestroy(plan->ev_start);
    if (plan->ev_stop) (void)hipEventDestroy(plan->ev_stop);
    delete plan;
}

int vlr_plan_n_out(const vlr_plan* plan) { return plan ? plan->n_events + 2 : VLR_ERR_INVALID_ARGUMENT; }
int vlr_plan_n_samples(const vlr_plan* plan) { return plan ? plan->host.S : VLR_ERR_INVALID_ARGUMENT; }

// LDS budget knob: maximum pileup depth per sample the kernel reserves coefficient space for
// (reference default max_depth = 200, src/variants/sample.rs:236).  Loci above it get VLR_LOCUS_TOO_DEEP.
int vlr_plan_set_max_depth(vlr_plan* plan, int per_sample_depth) {
    if (!plan || per_sample_depth < 1) return fail(VLR_ERR_INVALID_ARGUMENT, "invalid max depth");
    size_t lds = (size_t)2 * per_sample_depth * plan->host.S * 8;
    if (lds > 120 * 1024) return fail(VLR_ERR_UNSUPPORTED, "max depth %d x %d samples exceeds the LDS budget", per_sample_depth, plan->host.S);
    plan->max_depth_per_sample = per_sample_depth;
    plan->max_obs = 0;
    return VLR_OK;
}

// finer LDS knob: maximum number of kept observations of one locus over all samples
int vlr_plan_set_max_obs(vlr_plan* plan, int max_obs_per_locus) {
    if (!plan || max_obs_per_locus < 1) return fail(VLR_ERR_INVALID_ARGUMENT, "invalid max obs");
    if ((size_t)2 * max_obs_per_locus * 8 > 120 * 1024) return fail(VLR_ERR_UNSUPPORTED, "max obs %d exceeds the LDS budget", max_obs_per_locus);
    plan->max_obs = max_obs_per_locus;
    return VLR_OK;
}

// LDS budget for a batch whose pileups the caller knows (host offsets): the deepest locus — or, where a slightly smaller budget lets
// SIXTEEN workgroups share a CU (the launcher then runs the 4-wave build: +11..13 % on shallow tumor-normal batches,
// tools/waves4_lds_probe.py), at most 0.5 % of the loci exceed it and the batch is large enough to pay for the deep launch behind the
// call launch (one locus' latency, ~1 ms: tools/budget_probe.py), that budget.
static int fit_budget(vlr_plan* plan, const uint32_t* off, int64_t l0, int64_t l1, int S, int cap_budget) {
    uint32_t mx = 1;
    for (int64_t l = l0; l < l1; ++l) mx = std::max(mx, off[(l + 1) * S] - off[l * S]);
    int best = std::min<int>(cap_budget, (int)mx);
    if (plan->wide || getenv("VLR_NO_FIT_BUDGET")) return best;
    if (plan->lds_b0 == -2) {
        plan->lds_b0 = vlr_launch_call_lds_bytes(&plan->host8, plan->host.n_univ, S, 0, plan->host.max_range_depth);
        if (plan->lds_b0 < 0) (void)hipGetLastError();
    }
    const long long b0 = plan->lds_b0;
    if (b0 < 0) return best;
    if (b0 + 16ll * ((best + 3) & ~3) <= vlr::kLdsWg16 || b0 + 64 > vlr::kLdsWg16) return best;
    const int m16 = (int)((vlr::kLdsWg16 - b0) / 16) & ~3;   // largest budget with 16 workgroups per CU (the launch rounds budgets up to a multiple of four)
    if (m16 < 1 || m16 >= best) return best;
    int64_t over = 0;
    for (int64_t l = l0; l < l1; ++l) over += (off[(l + 1) * S] - off[l * S]) > (uint32_t)m16;
    return (over * 200 <= (l1 - l0) && (over == 0 || l1 - l0 >= 100000)) ? m16 : best;
}

int vlr_plan_fit_max_obs(vlr_plan* plan, const uint32_t* obs_offset_host, int64_t n_loci) {
    if (!plan || !obs_offset_host || n_loci < 0) return fail(VLR_ERR_INVALID_ARGUMENT, "invalid argument");
    int dev_before = 0;
    (void)hipGetDevice(&dev_before);
    (void)hipSetDevice(plan->device);
    const int S = plan->host.S;
    const int b = fit_budget(plan, obs_offset_host, 0, n_loci, S, 7680);
    (void)hipSetDevice(dev_before);
    const int rc = vlr_plan_set_max_obs(plan, std::max(1, b));
    return rc != VLR_OK ? rc : std::max(1, b);
}

// Device buffers a batch of n_loci needs besides the caller's: kernel scratch (third coefficients), and with AFD the replay
// scratch and the AFD log.  Grown here (hipMalloc/hipFree synchronise the device); vlr_batch_run calls this itself, callers that
// need a strictly asynchronous vlr_batch_run size the plan once with vlr_plan_reserve.
static int ensure_buffers(vlr_plan* plan, int64_t n_loci, int max_obs, bool want_afd, bool want_log, int afd_capacity) {
    const int k = plan->slot & 1;
    const size_t L = (size_t)n_loci;
    auto grow = [&](void** buf, size_t* have, size_t need, bool optional) -> int {
        if (need <= *have) return VLR_OK;
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr; *have = 0;
        if (hipMalloc(buf, need) != hipSuccess) {
            (void)hipGetLastError();
            *buf = nullptr;
            return optional ? VLR_OK : fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", need);
        }
        *have = need;
        return VLR_OK;
    };
    // (+ 2 S doubles per locus in front of every row: the all-ones products, DevResults::escratch)
    int rc = grow(&plan->escratch[k], &plan->escratch_bytes[k], L * ((size_t)max_obs + 2 * (size_t)plan->host.S) * sizeof(double), false);
    if (rc != VLR_OK) return rc;
    {   // pool of the deep launch: 24 B per kept observation of the loci above the LDS budget (default 512 MiB = 22 M observations
        // per batch; VLR_DEEP_POOL_MB, 0 = no deep launch: such loci stay flagged VLR_LOCUS_TOO_DEEP).  Optional: no room, no fallback.
        size_t pool = (size_t)512 << 20;
        if (const char* ev = getenv("VLR_DEEP_POOL_MB")) pool = (size_t)std::max(0L, atol(ev)) << 20;
        if (pool > 0 && plan->deep_hint != 0) (void)grow(&plan->deep_pool[k], &plan->deep_pool_bytes[k], pool + 128, true);
    }
    if (want_afd) {
        rc = grow(&plan->afd_scratch[k], &plan->afd_scratch_bytes[k], 2 * L + 8 * L + 4 * L + 64, false);
        if (rc != VLR_OK) return rc;
        if (want_log) {
            size_t words = 1 + (size_t)36 * (1 + plan->host.S + 2 * (size_t)plan->host.table_cap);
            words = std::min<size_t>((words + 63) & ~(size_t)63, (size_t)1 << 15);
            plan->afd_log_words = words;
            // budget (default 4 GiB, VLR_AFD_LOG_BUDGET_MB): a 1 M-locus tumor-normal batch would otherwise ask for 38 GB of
            // log; vlr_batch_run walks the batch in sub-ranges of afd_log_loci loci that share the buffer
            size_t budget = (size_t)4 << 30;
            if (const char* ev = getenv("VLR_AFD_LOG_BUDGET_MB")) budget = (size_t)std::max(1L, atol(ev)) << 20;
            size_t loci = std::max<size_t>(std::min<size_t>(L, budget / (words * sizeof(double))), std::min<size_t>(L, getenv("VLR_AFD_LOG_BUDGET_MB") ? 64 : 4096));
            if (plan->afd_log_loci[k] > 0 && plan->afd_log[k] && (size_t)plan->afd_log_loci[k] * words * sizeof(double) <= plan->afd_log_bytes[k])
                loci = std::max<size_t>(loci, std::min<size_t>(L, (size_t)plan->afd_log_loci[k]));
            (void)grow(&plan->afd_log[k], &plan->afd_log_bytes[k], loci * words * sizeof(double), true);  // no room: replay alone
            plan->afd_log_loci[k] = plan->afd_log[k] ? (int64_t)(plan->afd_log_bytes[k] / (words * sizeof(double))) : 0;
        } else {
            plan->afd_log_loci[k] = 0;
        }
        // one 8-byte key per AFD entry of the batch (as large as the caller's own afd_vaf): the l2fc part of the reference's map key
        rc = grow(&plan->afd_keys[k], &plan->afd_keys_bytes[k], L * (size_t)plan->host.S * (size_t)std::max(afd_capacity, 1) * sizeof(long long), false);
        if (rc != VLR_OK) return rc;
    }
    return VLR_OK;
}

int vlr_plan_reserve(vlr_plan* plan, int64_t n_loci, int with_afd) {
    if (!plan || n_loci < 0) return fail(VLR_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(plan->device));
    int max_obs = plan->max_obs > 0 ? plan->max_obs : plan->max_depth_per_sample * plan->host.S;
    max_obs = (max_obs + 3) & ~3;
    // both staging slots (vlr_batch_run_host alternates between them)
    const int slot0 = plan->slot;
    int rc = VLR_OK;
    for (int k = 0; k < 2 && rc == VLR_OK; ++k) {
        plan->slot = slot0 ^ k;
        rc = ensure_buffers(plan, n_loci, max_obs, with_afd != 0, with_afd != 0 && !getenv("VLR_AFD_REPLAY"), with_afd);
    }
    plan->slot = slot0;
    return rc;
}

int vlr_batch_run(vlr_plan* plan, const vlr_batch* in, vlr_results* out, void* stream) {
    using namespace vlr;
    if (!plan || !in || !out) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (in->n_samples != plan->host.S) return fail(VLR_ERR_INVALID_ARGUMENT, "batch has %d samples, plan %d", in->n_samples, plan->host.S);
    if (out->n_out != plan->n_events + 2 || out->n_samples != plan->host.S || out->n_loci < in->n_loci)
        return fail(VLR_ERR_INVALID_ARGUMENT, "result buffers do not match plan/batch");
    const bool want_afd = out->afd_count || out->afd_vaf || out->afd_lnprob;
    if (want_afd && (!out->afd_count || !out->afd_vaf || !out->afd_lnprob || out->afd_capacity < 1))
        return fail(VLR_ERR_INVALID_ARGUMENT, "AFD needs afd_count, afd_vaf, afd_lnprob and afd_capacity >= 1");
    if (!in->obs_offset || !in->prob_mapping || !in->prob_alt || !in->prob_ref || !in->prob_missed_allele || !in->prob_sample_alt ||
        !in->prob_double_overlap || !in->prob_hit_base || !in->flags || !in->locus_flags || !out->ln_posterior || !out->map_vaf || !out->status)
        return fail(VLR_ERR_INVALID_ARGUMENT, "missing required column");
    if (in->n_loci == 0) return VLR_OK;
    if (in->n_loci > 0x7fffffffLL) return fail(VLR_ERR_INVALID_ARGUMENT, "at most 2^31-1 loci per batch");
    HIP_TRY(hipSetDevice(plan->device));
    DevBatch b{};
    b.n_loci = in->n_loci;
    b.obs_offset = in->obs_offset;
    b.pm = in->prob_mapping; b.pa = in->prob_alt; b.pr = in->prob_ref; b.miss = in->prob_missed_allele;
    b.psa = in->prob_sample_alt; b.pdo = in->prob_double_overlap; b.phb = in->prob_hit_base;
    b.hpa = in->prob_hp_artifact; b.hpv = in->prob_hp_variant;
    b.flags = in->flags;
    b.locus_flags = in->locus_flags; b.variant_type = in->variant_type; b.ref_base = in->ref_base; b.alt_base = in->alt_base;
    DevResults r{};
    r.ln_posterior = out->ln_posterior; r.ln_marginal = out->ln_marginal; r.map_vaf = out->map_vaf;
    r.map_bias = out->map_bias; r.best_event = out->best_event; r.status = out->status;
    r.work = plan->work_dev;
    int max_obs = plan->max_obs > 0 ? plan->max_obs : plan->max_depth_per_sample * plan->host.S;
    max_obs = (max_obs + 3) & ~3;
    {
        const int rc0 = ensure_buffers(plan, in->n_loci, max_obs, want_afd, want_afd && !getenv("VLR_AFD_REPLAY"), want_afd ? out->afd_capacity : 0);
        if (rc0 != VLR_OK) return rc0;
    }
    if (want_afd) {
        // the replay pass needs MAP is_discrete flags, marginal and best event of the first pass
        const size_t L = (size_t)in->n_loci;
        const int k = plan->slot & 1;
        char* sc = (char*)plan->afd_scratch[k];
        if (!r.ln_marginal) r.ln_marginal = (double*)sc;
        if (!r.best_event) r.best_event = (int32_t*)(sc + 8 * L);
        r.map_disc = (uint16_t*)(sc + 12 * L);
        if (!r.map_bias) return fail(VLR_ERR_INVALID_ARGUMENT, "AFD needs map_bias");
        r.afd_count = out->afd_count; r.afd_vaf = out->afd_vaf; r.afd_lnprob = out->afd_lnprob; r.afd_capacity = out->afd_capacity;
        r.afd_key = (long long*)plan->afd_keys[k];
    }
    r.escratch = (double*)plan->escratch[plan->slot & 1];
    if (want_afd && !getenv("VLR_AFD_REPLAY") && plan->afd_log[plan->slot & 1]) {
        r.afd_log = (double*)plan->afd_log[plan->slot & 1];
        r.afd_log_stride = (long long)plan->afd_log_words;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipEventRecord(plan->ev_start, st));
    // plans beyond the standard build's limits (vlr_plan.h) run the wide build of the kernels (vlr_kernels_wide.hip)
    const bool wide = plan->wide;
    auto call_launch = [&](const DevBatch* bb, const DevResults* rr, int mo, void* ss) {
        return wide ? vlr_launch_call_kernel_wide(&plan->host, bb, rr, plan->host.n_univ, plan->host.S, mo, plan->host.max_range_depth, ss)
                    : vlr_launch_call_kernel(&plan->host8, bb, rr, plan->host.n_univ, plan->host.S, mo, plan->host.max_range_depth, ss);
    };
    auto afd_launch = [&](const DevBatch* bb, const DevResults* rr, void* ss) {
        return wide ? vlr_launch_afd_kernel_wide(&plan->host, bb, rr, ss) : vlr_launch_afd_kernel(&plan->host8, bb, rr, ss);
    };
    r.deep_used = (unsigned long long*)plan->deep_pool[plan->slot & 1];  // reset by workgroup 0 of every LDS-resident launch
    // deep launch behind a call (or replay) launch: re-evaluates the loci that launch flagged VLR_LOCUS_TOO_DEEP with their
    // coefficients in the plan's HBM pool; every other locus exits at once.  `lane`/`two`: the AFD sub-range lanes share the pool.
    auto deep_launch = [&](const DevBatch& bs, DevResults rs, void* ss, int lane, bool two) -> int {
        const int k = plan->slot & 1;
        if (!plan->deep_pool[k] || plan->deep_pool_bytes[k] <= 128 || plan->deep_hint == 0) return VLR_OK;
        char* base = (char*)plan->deep_pool[k];
        unsigned long long* ctr = (unsigned long long*)(base + 64 * lane);
        size_t cap_d = (plan->deep_pool_bytes[k] - 128) / sizeof(double);
        double* data = (double*)(base + 128);
        if (two) { cap_d /= 2; data += (size_t)lane * cap_d; }
        rs.deep_pool = data; rs.deep_used = ctr; rs.deep_capacity = (long long)cap_d;  // ctr was reset by the launch before (workgroup 0)
        const int rc = wide ? vlr_launch_call_kernel_widedeep(&plan->host, &bs, &rs, plan->host.n_univ, plan->host.S, plan->host.max_range_depth, ss)
                            : vlr_launch_call_kernel_deep(&plan->host8, &bs, &rs, plan->host.n_univ, plan->host.S, plan->host.max_range_depth, ss);
        if (rc != 0) return fail(VLR_ERR_HIP, "deep kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        return VLR_OK;
    };
    if (!want_afd) {
        int rc = call_launch(&b, &r, max_obs, stream);
        if (rc != 0) return fail(VLR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        rc = deep_launch(b, r, stream, 0, false);
        if (rc != VLR_OK) return rc;
        HIP_TRY(hipEventRecord(plan->ev_stop, st));
    } else {
        // FORMAT/AFD (calling.rs:889-928): from the log of the call pass; replay of the clean events where the log overflowed.
        // The log region of a locus is sized for the worst case (tens of kB), so the batch is walked in sub-ranges of loci whose
        // logs fit the plan's budget (afd_log_loci, ensure_buffers) and reuse one buffer: call pass, log filter and replay of a
        // sub-range are stream-ordered one behind the other.
        HIP_TRY(hipMemsetAsync(out->afd_count, 0, (size_t)in->n_loci * plan->host.S * sizeof(int32_t), st));
        const int S = plan->host.S, n_out = out->n_out;
        const int64_t cap = (r.afd_log && plan->afd_log_loci[plan->slot & 1] > 0) ? plan->afd_log_loci[plan->slot & 1] : in->n_loci;
        // one sub-range when everything fits; otherwise two lanes (the caller's stream and a plan-owned one, each with half of the
        // log) so that the long tail of one sub-range — a few nested loci cost ten times the average — overlaps the next
        const bool two = in->n_loci > cap && cap >= 2;
        const int64_t step = two ? cap / 2 : std::max<int64_t>(cap, 1);
        if (two) {
            if (!plan->afd_aux_stream) {
                HIP_TRY(hipStreamCreateWithFlags(&plan->afd_aux_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&plan->ev_join, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(plan->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(plan->afd_aux_stream, plan->ev_fork, 0));
        }
        int64_t k = 0;
        for (int64_t l0 = 0; l0 < in->n_loci; l0 += step, ++k) {
            const int lane = two ? (int)(k & 1) : 0;
            void* ss = lane ? (void*)plan->afd_aux_stream : stream;
            DevBatch bs = b;
            DevResults rs = r;
            bs.n_loci = std::min<int64_t>(step, in->n_loci - l0);
            bs.obs_offset += l0 * S;
            bs.locus_flags += l0;
            if (bs.variant_type) bs.variant_type += l0;
            if (bs.ref_base) bs.ref_base += l0;
            if (bs.alt_base) bs.alt_base += l0;
            rs.ln_posterior += l0 * n_out; rs.ln_marginal += l0; rs.map_vaf += l0 * S; rs.map_bias += l0 * 6; rs.best_event += l0;
            rs.status += l0; rs.map_disc += l0; rs.escratch += (size_t)l0 * ((size_t)max_obs + 2 * (size_t)S);
            rs.afd_count += l0 * S; rs.afd_vaf += (size_t)l0 * S * r.afd_capacity; rs.afd_lnprob += (size_t)l0 * S * r.afd_capacity;
            if (rs.afd_log) rs.afd_log += (size_t)lane * (size_t)step * (size_t)r.afd_log_stride;
            rs.afd_key += (size_t)l0 * (size_t)S * (size_t)r.afd_capacity;
            int rc = call_launch(&bs, &rs, max_obs, ss);
            if (rc != 0) return fail(VLR_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            if (rs.afd_log) {
                rc = afd_launch(&bs, &rs, ss);
                if (rc != 0) return fail(VLR_ERR_HIP, "AFD kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            }
            rs.replay = 1;
            rc = call_launch(&bs, &rs, max_obs, ss);
            if (rc != 0) return fail(VLR_ERR_HIP, "AFD replay launch failed: %s", hipGetErrorString((hipError_t)rc));

        }
        if (two) {
            HIP_TRY(hipEventRecord(plan->ev_join, plan->afd_aux_stream));
            HIP_TRY(hipStreamWaitEvent(st, plan->ev_join, 0));
        }
        // the loci above the LDS budget, once for the whole batch: call launch, then (counter reset in between) the lists
        {
            DevResults rd = r;
            int rc = deep_launch(b, rd, stream, 0, false);
            if (rc != VLR_OK) return rc;
            if (plan->deep_pool[plan->slot & 1] && plan->deep_hint != 0) {
                HIP_TRY(hipMemsetAsync(plan->deep_pool[plan->slot & 1], 0, sizeof(unsigned long long), st));
                rd.replay = 1;
                rc = deep_launch(b, rd, stream, 0, false);
                if (rc != VLR_OK) return rc;
            }
        }
        HIP_TRY(hipEventRecord(plan->ev_stop, st));
    }
    plan->timed = true;
    return VLR_OK;
}

int vlr_plan_last_kernel_ms(vlr_plan* plan, float* ms) {
    if (!plan || !ms) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (!plan->timed) return fail(VLR_ERR_INVALID_ARGUMENT, "no batch has been run on this plan");
    HIP_TRY(hipEventSynchronize(plan->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, plan->ev_start, plan->ev_stop));
    return VLR_OK;
}

// profiling aid: cumulative {pileup evaluations, observation terms} since plan creation (synchronises)
int vlr_plan_work_counters(vlr_plan* plan, unsigned long long* out2, int reset) {
    if (!plan || !out2) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out2, plan->work_dev, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(plan->work_dev, 0, 64 * sizeof(unsigned long long)));
    return VLR_OK;
}

// developer aid: per-phase wave cycles (only meaningful in a -DVLR_PROFILE build); 24 counters
extern "C" int vlr_plan_profile_counters(vlr_plan* plan, unsigned long long* out12) {
    if (!plan || !out12) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out12, plan->work_dev + 2, 40 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VLR_OK;
}

// One chunk of loci [l0, l1) of a host batch: stage the inputs in slot k, launch on the slot's stream.  The result
// copies are issued later by host_chunk_finish so that they do not stall the staging of the next chunk.
struct HostChunk {
    int64_t l0 = 0, l1 = 0;
    int k = 0;
    vlr_results dr{};
    bool want_afd = false;
    bool active = false;
    // FORMAT/AFD text of the chunk (vlr_results.afd_text): device staging and the chunk's share of the caller's text buffer
    uint8_t* text_dev = nullptr;
    uint32_t* span_dev = nullptr;
    uint32_t* cur_dev = nullptr;
    uint16_t* rank_dev = nullptr;
    uint64_t text_cap = 0, text_base = 0;
};

// (dev_off != nullptr: the columns of `in` are DEVICE arrays already — the batch of a device reader — and dev_off is the host copy of
// its obs_offset; nothing is staged but the result buffers)
static int host_chunk_start(vlr_plan* plan, const vlr_batch* in, vlr_results* out, int64_t l0, int64_t l1, int k, HostChunk* hc, const uint32_t* dev_off = nullptr) {
    const int S = plan->host.S;
    const int64_t L = l1 - l0;
    const int n_out = plan->n_events + 2;
    const uint32_t* h_off = dev_off ? dev_off : in->obs_offset;
    const uint32_t ob = h_off[l0 * S], oe = h_off[l1 * S];
    const size_t N = (size_t)(oe - ob);
    hipStream_t st = plan->stage_stream[k];
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    struct Col { const void* src; size_t bytes; size_t off; };
    std::vector<Col> cols;
    size_t off = 0;
    auto add = [&](const void* p, size_t bytes) {
        if (dev_off) return (size_t)0;
        cols.push_back({p, bytes, off});
        size_t o = off;
        off += al(std::max<size_t>(bytes, 1));
        return o;
    };
    auto obs = [&](const float* p) { return p ? (const void*)(p + ob) : nullptr; };
    size_t o_off = add(in->obs_offset + l0 * S, (size_t)(L * S + 1) * 4);
    size_t o_pm = add(obs(in->prob_mapping), N * 4), o_pa = add(obs(in->prob_alt), N * 4), o_pr = add(obs(in->prob_ref), N * 4);
    size_t o_ms = add(obs(in->prob_missed_allele), N * 4), o_psa = add(obs(in->prob_sample_alt), N * 4), o_pdo = add(obs(in->prob_double_overlap), N * 4);
    size_t o_phb = add(obs(in->prob_hit_base), N * 4);
    size_t o_hpa = add(obs(in->prob_hp_artifact), in->prob_hp_artifact ? N * 4 : 0), o_hpv = add(obs(in->prob_hp_variant), in->prob_hp_variant ? N * 4 : 0);
    size_t o_fl = add(in->flags + ob, N * 4);
    size_t o_lf = add(in->locus_flags + l0, L), o_vt = add(in->variant_type ? in->variant_type + l0 : nullptr, in->variant_type ? L : 0);
    size_t o_rb = add(in->ref_base ? in->ref_base + l0 : nullptr, in->ref_base ? L : 0), o_ab = add(in->alt_base ? in->alt_base + l0 : nullptr, in->alt_base ? L : 0);
    size_t r_post = off; off += al((size_t)L * n_out * 8);
    size_t r_marg = off; off += al((size_t)L * 8);
    size_t r_map = off; off += al((size_t)L * S * 8);
    size_t r_bias = off; off += al((size_t)L * VLR_N_BIAS);
    size_t r_best = off; off += al((size_t)L * 4);
    size_t r_stat = off; off += al((size_t)L * 4);
    const bool want_afd = out->afd_count || out->afd_vaf || out->afd_lnprob;
    const size_t cap = want_afd ? (size_t)out->afd_capacity : 0;
    size_t r_ac = off; off += al(want_afd ? (size_t)L * S * 4 : 0);
    size_t r_av = off; off += al((size_t)L * S * cap * 8);
    size_t r_al = off; off += al((size_t)L * S * cap * 8);
    // FORMAT/AFD text (vlr_results.afd_text): the chunk formats into its share [cap l0 / n, cap l1 / n) of the caller's buffer
    const bool want_text = want_afd && out->afd_text && out->afd_text_span && out->afd_text_capacity >= 16;
    uint64_t text_base = 0, text_cap = 0;
    if (want_text) {
        const uint64_t tot = std::min<uint64_t>(out->afd_text_capacity, 0xffffff00ull);
        text_base = (uint64_t)((double)tot * (double)l0 / (double)in->n_loci) & ~15ull;
        const uint64_t text_end = l1 >= in->n_loci ? tot : ((uint64_t)((double)tot * (double)l1 / (double)in->n_loci) & ~15ull);
        text_cap = text_end > text_base ? text_end - text_base : 0;
    }
    size_t r_tx = off; off += al(want_text ? (size_t)text_cap : 0);
    size_t r_sp = off; off += al(want_text ? (size_t)L * S * 8 : 0);
    size_t r_cu = off; off += al(want_text ? 64 : 0);
    size_t r_rk = off; off += al(want_text ? (size_t)L * S * cap * 2 : 0);
    if (off > plan->stage_bytes[k]) {
        if (plan->stage[k]) (void)hipFree(plan->stage[k]);
        plan->stage[k] = nullptr;
        plan->stage_bytes[k] = 0;
        if (hipMalloc(&plan->stage[k], off) != hipSuccess) return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed", off);
        plan->stage_bytes[k] = off;
    }
    char* base = (char*)plan->stage[k];
    // (A pinned bounce buffer filled by CPU threads was measured slower than the runtime's own pageable path on the
    //  test box, 15 vs 14 GB/s with more CPU; callers that can hand over pinned arrays get true async DMA here.)
    for (auto& c : cols)
        if (c.src && c.bytes) HIP_TRY(hipMemcpyAsync(base + c.off, c.src, c.bytes, hipMemcpyHostToDevice, st));
    // the kernel indexes the observation columns with the absolute row numbers of obs_offset: bias the column pointers
    vlr_batch db = *in;
    db.n_loci = L;
    db.n_obs = (int64_t)N;
    db.obs_offset = (const uint32_t*)(base + o_off);
    auto colp = [&](size_t o) { return (const float*)(base + o) - ob; };
    db.prob_mapping = colp(o_pm); db.prob_alt = colp(o_pa); db.prob_ref = colp(o_pr);
    db.prob_missed_allele = colp(o_ms); db.prob_sample_alt = colp(o_psa);
    db.prob_double_overlap = colp(o_pdo); db.prob_hit_base = colp(o_phb);
    db.prob_hp_artifact = in->prob_hp_artifact ? colp(o_hpa) : nullptr;
    db.prob_hp_variant = in->prob_hp_variant ? colp(o_hpv) : nullptr;
    db.flags = (const uint32_t*)(base + o_fl) - ob;
    db.locus_flags = (const uint8_t*)(base + o_lf);
    db.variant_type = in->variant_type ? (const uint8_t*)(base + o_vt) : nullptr;
    db.ref_base = in->ref_base ? (const uint8_t*)(base + o_rb) : nullptr;
    db.alt_base = in->alt_base ? (const uint8_t*)(base + o_ab) : nullptr;
    if (dev_off) {   // the caller's device arrays: observation columns are indexed absolutely, the per-locus ones advance
        db = *in;
        db.n_loci = L; db.n_obs = (int64_t)N;
        db.obs_offset = in->obs_offset + l0 * S;
        db.locus_flags = in->locus_flags + l0;
        db.variant_type = in->variant_type ? in->variant_type + l0 : nullptr;
        db.ref_base = in->ref_base ? in->ref_base + l0 : nullptr;
        db.alt_base = in->alt_base ? in->alt_base + l0 : nullptr;
    }
    vlr_results dr = *out;
    dr.n_loci = L;
    dr.ln_posterior = (double*)(base + r_post);
    dr.ln_marginal = (double*)(base + r_marg);
    dr.map_vaf = (double*)(base + r_map);
    dr.map_bias = (uint8_t*)(base + r_bias);
    dr.best_event = (int32_t*)(base + r_best);
    dr.status = (uint32_t*)(base + r_stat);
    dr.afd_count = want_afd ? (int32_t*)(base + r_ac) : nullptr;
    dr.afd_vaf = want_afd ? (double*)(base + r_av) : nullptr;
    dr.afd_lnprob = want_afd ? (double*)(base + r_al) : nullptr;
    // size the LDS coefficient area to this chunk (never above the configured budget)
    int saved_max_obs = plan->max_obs;
    {
        int budget = plan->max_obs > 0 ? plan->max_obs : plan->max_depth_per_sample * S;
        uint32_t mx = 1;
        for (int64_t l = l0; l < l1; ++l) mx = std::max(mx, h_off[(l + 1) * S] - h_off[l * S]);
        plan->max_obs = fit_budget(plan, h_off, l0, l1, S, budget);   // (the deepest locus, or the 16-workgroup budget: see fit_budget)
        // the LDS-resident kernel holds at most 7 680 kept observations of a locus; the launcher's own limit is the budget
        plan->deep_hint = ((int)mx > std::min(plan->max_obs, 7680)) ? 1 : 0;
    }
    plan->slot = k;
    int rc = vlr_batch_run(plan, &db, &dr, (void*)st);
    plan->max_obs = saved_max_obs;
    plan->deep_hint = -1;
    if (rc != VLR_OK) return rc;
    hc->text_dev = nullptr;
    if (want_text) {   // behind the AFD passes on the chunk's stream: the lists are final in the staging arrays
        hc->text_dev = (uint8_t*)(base + r_tx); hc->span_dev = (uint32_t*)(base + r_sp); hc->cur_dev = (uint32_t*)(base + r_cu);
        hc->rank_dev = (uint16_t*)(base + r_rk);
        hc->text_cap = text_cap; hc->text_base = text_base;
        rc = vlr_launch_afd_text(dr.afd_count, dr.afd_vaf, dr.afd_lnprob, L * S, (int)cap, hc->text_dev, (uint32_t)text_cap, hc->span_dev, hc->rank_dev, hc->cur_dev, (void*)st);
        if (rc != VLR_OK) return fail(rc, "AFD text kernel launch failed");
    }
    hc->l0 = l0; hc->l1 = l1; hc->k = k; hc->dr = dr; hc->want_afd = want_afd; hc->active = true;
    return VLR_OK;
}

static int host_chunk_finish(vlr_plan* plan, vlr_results* out, HostChunk* hc) {
    if (!hc->active) return VLR_OK;
    const int S = plan->host.S;
    const int n_out = plan->n_events + 2;
    const int64_t l0 = hc->l0, L = hc->l1 - hc->l0;
    const size_t cap = hc->want_afd ? (size_t)out->afd_capacity : 0;
    hipStream_t st = plan->stage_stream[hc->k];
    const vlr_results& dr = hc->dr;
    HIP_TRY(hipMemcpyAsync(out->ln_posterior + l0 * n_out, dr.ln_posterior, (size_t)L * n_out * 8, hipMemcpyDeviceToHost, st));
    if (out->ln_marginal) HIP_TRY(hipMemcpyAsync(out->ln_marginal + l0, dr.ln_marginal, (size_t)L * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->map_vaf + l0 * S, dr.map_vaf, (size_t)L * S * 8, hipMemcpyDeviceToHost, st));
    if (out->map_bias) HIP_TRY(hipMemcpyAsync(out->map_bias + l0 * VLR_N_BIAS, dr.map_bias, (size_t)L * VLR_N_BIAS, hipMemcpyDeviceToHost, st));
    if (out->best_event) HIP_TRY(hipMemcpyAsync(out->best_event + l0, dr.best_event, (size_t)L * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out->status + l0, dr.status, (size_t)L * 4, hipMemcpyDeviceToHost, st));
    if (hc->want_afd) {
        HIP_TRY(hipMemcpyAsync(out->afd_count + l0 * S, dr.afd_count, (size_t)L * S * 4, hipMemcpyDeviceToHost, st));
        bool lists = true;
        if (hc->text_dev) {
            // the text instead of the lists: 12 bytes per ENTRY against 16 per SLOT of the capacity; the lists only follow when the
            // kernel left one of them to the host
            uint32_t cur[4] = {0, 0, 0, 0};
            uint32_t* sp = out->afd_text_span + (size_t)l0 * S * 2;
            HIP_TRY(hipMemcpyAsync(cur, hc->cur_dev, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(sp, hc->span_dev, (size_t)L * S * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const size_t used = (size_t)std::min<uint64_t>(cur[0], hc->text_cap);
            if (used) HIP_TRY(hipMemcpyAsync(out->afd_text + hc->text_base, hc->text_dev, used, hipMemcpyDeviceToHost, st));
            lists = cur[2] != 0;
            if (hc->text_base)
                for (size_t q = 0; q < (size_t)L * S; ++q)
                    if (sp[2 * q + 1] != 0xffffffffu) sp[2 * q] += (uint32_t)hc->text_base;
        }
        if (lists) {
            HIP_TRY(hipMemcpyAsync(out->afd_vaf + l0 * S * cap, dr.afd_vaf, (size_t)L * S * cap * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(out->afd_lnprob + l0 * S * cap, dr.afd_lnprob, (size_t)L * S * cap * 8, hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    hc->active = false;
    return VLR_OK;
}

// Host buffers in, host buffers out.  The loci are cut into chunks of observation data (size rule below); chunk c+1 is staged
// (H2D) while the kernel of chunk c runs on the other slot's stream, and the results of chunk c are fetched after the
// launch of chunk c+1 — the PCIe transfers disappear behind the kernels.
int vlr_batch_run_host(vlr_plan* plan, const vlr_batch* in, vlr_results* out) {
    if (!plan || !in || !out) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    const int S = plan->host.S;
    const int64_t L = in->n_loci;
    if (L == 0) return VLR_OK;
    if (in->n_samples != S) return fail(VLR_ERR_INVALID_ARGUMENT, "batch has %d samples, plan %d", in->n_samples, S);
    if (!in->obs_offset || !in->flags || !in->locus_flags) return fail(VLR_ERR_INVALID_ARGUMENT, "missing required column");
    if (in->n_obs > 0 && (!in->prob_mapping || !in->prob_alt || !in->prob_ref || !in->prob_missed_allele || !in->prob_sample_alt ||
                          !in->prob_double_overlap || !in->prob_hit_base))
        return fail(VLR_ERR_INVALID_ARGUMENT, "missing required probability column (only the homopolymer columns are optional)");
    if (in->n_obs < 0 || in->n_obs > 0xffffffffll) return fail(VLR_ERR_INVALID_ARGUMENT, "n_obs %lld does not fit the 32-bit observation offsets", (long long)in->n_obs);
    const bool want_afd = out->afd_count || out->afd_vaf || out->afd_lnprob;
    if (want_afd && (!out->afd_count || !out->afd_vaf || !out->afd_lnprob || out->afd_capacity < 1))
        return fail(VLR_ERR_INVALID_ARGUMENT, "AFD needs afd_count, afd_vaf, afd_lnprob and afd_capacity >= 1");
    for (int k = 0; k < 2; ++k)
        if (!plan->stage_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&plan->stage_stream[k], hipStreamNonBlocking));
    // chunk boundaries: by bytes of observation columns (40 B per observation), at least 8192 loci per chunk
    const size_t bytes_total = (size_t)in->n_obs * 40 + (size_t)L * (S + 1) * 4;
    // Every chunk costs about a millisecond (launch tail of a small grid + the staging calls) and the transfer of the
    // first chunk is exposed: with ~14 GB/s from pageable memory, n = sqrt(transfer time in ms) chunks minimise the sum
    // (8 GB -> 24 chunks of ~330 MB; measured 0.39 s against 0.50 s with 64 MB chunks and 0.355 s for the kernel alone)
    int64_t n_chunks = std::max<int64_t>(1, (int64_t)std::sqrt((double)bytes_total / 14.0e6));
    if (const char* e = getenv("VLR_HOST_CHUNK_MB"))  // tuning knob
        n_chunks = (int64_t)std::max<size_t>(1, bytes_total / (std::max<size_t>(1, (size_t)atol(e)) << 20));
    n_chunks = std::min<int64_t>(n_chunks, std::max<int64_t>(1, L / 8192));
    HostChunk hc[2];
    int rc = VLR_OK;
    for (int64_t c = 0; c < n_chunks && rc == VLR_OK; ++c) {
        const int k = (int)(c & 1);
        const int64_t l0 = L * c / n_chunks, l1 = L * (c + 1) / n_chunks;
        rc = host_chunk_start(plan, in, out, l0, l1, k, &hc[k]);
        if (rc == VLR_OK) rc = host_chunk_finish(plan, out, &hc[k ^ 1]);  // previous chunk: overlaps the kernel just launched
    }
    for (int k = 0; k < 2; ++k) {
        int r2 = host_chunk_finish(plan, out, &hc[k]);
        if (rc == VLR_OK) rc = r2;
    }
    if (rc != VLR_OK) (void)hipDeviceSynchronize();
    return rc;
}

// Device columns in (the batch of a device reader, vlr_obs_table_device_batch), host results out: the evaluation stage of the
// front door when the observation columns were decoded on the device.  obs_offset_host: host copy of in->obs_offset.
int vlr_batch_run_device_in(vlr_plan* plan, const vlr_batch* in, const uint32_t* obs_offset_host, vlr_results* out) {
    if (!plan || !in || !out || !obs_offset_host) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    const int S = plan->host.S;
    const int64_t L = in->n_loci;
    if (L == 0) return VLR_OK;
    if (in->n_samples != S) return fail(VLR_ERR_INVALID_ARGUMENT, "batch has %d samples, plan %d", in->n_samples, S);
    if (!in->obs_offset || !in->flags || !in->locus_flags) return fail(VLR_ERR_INVALID_ARGUMENT, "missing required column");
    const bool want_afd = out->afd_count || out->afd_vaf || out->afd_lnprob;
    if (want_afd && (!out->afd_count || !out->afd_vaf || !out->afd_lnprob || out->afd_capacity < 1))
        return fail(VLR_ERR_INVALID_ARGUMENT, "AFD needs afd_count, afd_vaf, afd_lnprob and afd_capacity >= 1");
    for (int k = 0; k < 2; ++k)
        if (!plan->stage_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&plan->stage_stream[k], hipStreamNonBlocking));
    // chunks only bound the result staging (the AFD lists are 1.5 kB per locus and sample at capacity 96) and let the copy of one
    // chunk's results run behind the kernel of the next
    const size_t per_locus = (size_t)(plan->n_events + 2 + S + 1) * 8 + 16 + (want_afd ? (size_t)S * (4 + 16 * (size_t)out->afd_capacity) : 0);
    int64_t n_chunks = std::max<int64_t>(1, (int64_t)((double)per_locus * (double)L / 256.0e6));
    n_chunks = std::min<int64_t>(n_chunks, std::max<int64_t>(1, L / 8192));
    HostChunk hc[2];
    int rc = VLR_OK;
    for (int64_t c = 0; c < n_chunks && rc == VLR_OK; ++c) {
        const int k = (int)(c & 1);
        const int64_t l0 = L * c / n_chunks, l1 = L * (c + 1) / n_chunks;
        rc = host_chunk_start(plan, in, out, l0, l1, k, &hc[k], obs_offset_host);
        if (rc == VLR_OK) rc = host_chunk_finish(plan, out, &hc[k ^ 1]);
    }
    for (int k = 0; k < 2; ++k) {
        int r2 = host_chunk_finish(plan, out, &hc[k]);
        if (rc == VLR_OK) rc = r2;
    }
    if (rc != VLR_OK) (void)hipDeviceSynchronize();
    return rc;
}

// ---- one node, several devices (SURVEY 8 e): contiguous shards of one host batch, one thread + plan per device -------------
// The per-record loop of Caller::call (calling.rs:320-455) has no cross-record state: given the plan every locus is independent,
// so N devices take the blocks [n r / N, n (r + 1) / N) of the (breakend-group de-duplicated, calling.rs:569-580) batch and write
// their fixed-size records and AFD lists straight into the caller's arrays at the shard's offset — input order is restored by
// construction and no collective is needed inside one process (the multi-process harness uses one RCCL all-gather instead).
struct vlr_gpu_node {
    std::vector<vlr_plan*> plans;
    std::vector<int> devices;
};

int vlr_node_shard_range(int64_t n_loci, int n_shards, int shard, int64_t* l0, int64_t* l1) {
    if (n_loci < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !l0 || !l1) return fail(VLR_ERR_INVALID_ARGUMENT, "bad shard request");
    const int64_t per = (n_loci + n_shards - 1) / n_shards;  // ceil(n / G) loci per shard, the last ones may be short or empty
    *l0 = std::min<int64_t>(n_loci, per * shard);
    *l1 = std::min<int64_t>(n_loci, per * (shard + 1));
    return VLR_OK;
}

int vlr_node_create(const vlr_scenario_desc* d, int n_devices, const int* devices, vlr_gpu_node** out) {
    if (!d || !out) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); return fail(VLR_ERR_NO_DEVICE, "no HIP device (the engine has no CPU path)"); }
    if (n_devices <= 0) { n_devices = ndev; devices = nullptr; }
    auto* node = new vlr_gpu_node();
    for (int r = 0; r < n_devices; ++r) {
        const int dev = devices ? devices[r] : r;
        if (dev < 0 || dev >= ndev) { vlr_node_destroy(node); return fail(VLR_ERR_NO_DEVICE, "no HIP device %d (%d visible)", dev, ndev); }
        vlr_plan* p = nullptr;
        const int rc = vlr_plan_create(d, dev, &p);
        if (rc != VLR_OK) { vlr_node_destroy(node); return rc; }
        node->plans.push_back(p);
        node->devices.push_back(dev);
    }
    *out = node;
    return VLR_OK;
}

void vlr_node_destroy(vlr_gpu_node* node) {
    if (!node) return;
    for (vlr_plan* p : node->plans) vlr_plan_destroy(p);
    delete node;
}

int vlr_node_n_devices(const vlr_gpu_node* node) { return node ? (int)node->plans.size() : VLR_ERR_INVALID_ARGUMENT; }
int vlr_node_device(const vlr_gpu_node* node, int shard) {
    return (node && shard >= 0 && shard < (int)node->devices.size()) ? node->devices[(size_t)shard] : VLR_ERR_INVALID_ARGUMENT;
}
vlr_plan* vlr_node_plan(vlr_gpu_node* node, int shard) {
    return (node && shard >= 0 && shard < (int)node->plans.size()) ? node->plans[(size_t)shard] : nullptr;
}
int vlr_node_set_max_depth(vlr_gpu_node* node, int per_sample_depth) {
    if (!node) return fail(VLR_ERR_INVALID_ARGUMENT, "null node");
    for (vlr_plan* p : node->plans) { const int rc = vlr_plan_set_max_depth(p, per_sample_depth); if (rc != VLR_OK) return rc; }
    return VLR_OK;
}
int vlr_node_set_max_obs(vlr_gpu_node* node, int max_obs_per_locus) {
    if (!node) return fail(VLR_ERR_INVALID_ARGUMENT, "null node");
    for (vlr_plan* p : node->plans) { const int rc = vlr_plan_set_max_obs(p, max_obs_per_locus); if (rc != VLR_OK) return rc; }
    return VLR_OK;
}

int vlr_node_batch_run_host(vlr_gpu_node* node, const vlr_batch* in, vlr_results* out) {
    if (!node || !in || !out) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    const int G = (int)node->plans.size();
    if (G < 1) return fail(VLR_ERR_INVALID_ARGUMENT, "node without devices");
    const int64_t L = in->n_loci;
    if (L == 0) return VLR_OK;
    if (!in->obs_offset) return fail(VLR_ERR_INVALID_ARGUMENT, "missing required column");
    const int S = in->n_samples;
    const int n_out = vlr_plan_n_out(node->plans[0]);
    const bool want_afd = out->afd_count || out->afd_vaf || out->afd_lnprob;
    const size_t cap = want_afd ? (size_t)out->afd_capacity : 0;
    std::vector<int> rcs((size_t)G, VLR_OK);
    std::vector<std::string> errs((size_t)G);
    auto work = [&](int r) {
        int64_t l0 = 0, l1 = 0;
        (void)vlr_node_shard_range(L, G, r, &l0, &l1);
        if (l1 <= l0) return;
        // a view of the shard: the locus columns advance by l0, the observation columns stay (obs_offset carries absolute rows)
        vlr_batch b = *in;
        b.n_loci = l1 - l0;
        b.obs_offset = in->obs_offset + l0 * S;
        b.n_obs = (int64_t)in->obs_offset[l1 * S] - (int64_t)in->obs_offset[l0 * S];
        b.locus_flags = in->locus_flags ? in->locus_flags + l0 : nullptr;
        b.variant_type = in->variant_type ? in->variant_type + l0 : nullptr;
        b.ref_base = in->ref_base ? in->ref_base + l0 : nullptr;
        b.alt_base = in->alt_base ? in->alt_base + l0 : nullptr;
        vlr_results o = *out;
        o.n_loci = l1 - l0;
        o.ln_posterior = out->ln_posterior ? out->ln_posterior + l0 * n_out : nullptr;
        o.ln_marginal = out->ln_marginal ? out->ln_marginal + l0 : nullptr;
        o.map_vaf = out->map_vaf ? out->map_vaf + l0 * S : nullptr;
        o.map_bias = out->map_bias ? out->map_bias + l0 * VLR_N_BIAS : nullptr;
        o.best_event = out->best_event ? out->best_event + l0 : nullptr;
        o.status = out->status ? out->status + l0 : nullptr;
        o.afd_count = out->afd_count ? out->afd_count + l0 * S : nullptr;
        o.afd_vaf = out->afd_vaf ? out->afd_vaf + (size_t)l0 * S * cap : nullptr;
        o.afd_lnprob = out->afd_lnprob ? out->afd_lnprob + (size_t)l0 * S * cap : nullptr;
        uint64_t text_base = 0;
        if (out->afd_text && out->afd_text_span) {   // the shard's share of the text buffer; its spans are moved to the caller's offsets below
            const uint64_t tot = std::min<uint64_t>(out->afd_text_capacity, 0xffffff00ull);
            text_base = (uint64_t)((double)tot * (double)l0 / (double)L) & ~15ull;
            const uint64_t text_end = l1 >= L ? tot : ((uint64_t)((double)tot * (double)l1 / (double)L) & ~15ull);
            o.afd_text = out->afd_text + text_base;
            o.afd_text_capacity = text_end - text_base;
            o.afd_text_span = out->afd_text_span + (size_t)l0 * S * 2;
        }
        rcs[(size_t)r] = vlr_batch_run_host(node->plans[(size_t)r], &b, &o);
        if (rcs[(size_t)r] == VLR_OK && o.afd_text && text_base)
            for (size_t q = 0; q < (size_t)(l1 - l0) * S; ++q)
                if (o.afd_text_span[2 * q + 1] != 0xffffffffu) o.afd_text_span[2 * q] += (uint32_t)text_base;
        if (rcs[(size_t)r] != VLR_OK) errs[(size_t)r] = g_err;  // (thread-local: carried back to the caller's thread below)
    };
    if (G == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < G; ++r) th.emplace_back(work, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < G; ++r)
        if (rcs[(size_t)r] != VLR_OK) { g_err = "device " + std::to_string(node->devices[(size_t)r]) + ": " + errs[(size_t)r]; return rcs[(size_t)r]; }
    return VLR_OK;
}

// the device formatter of vlr_results.afd_text on host arrays (tests)
int vlr_selftest_afd_text(int device, int64_t n_lists, int capacity, const int32_t* count, const double* vaf, const double* lnprob, uint8_t* text,
                          uint64_t text_capacity, uint32_t* span, uint32_t* n_unformatted) {
    if (n_lists < 0 || capacity < 1 || !count || !vaf || !lnprob || !text || !span) return fail(VLR_ERR_INVALID_ARGUMENT, "bad argument");
    if (n_lists == 0) return VLR_OK;
    HIP_TRY(hipSetDevice(device));
    const size_t slots = (size_t)n_lists * (size_t)capacity, tcap = (size_t)std::min<uint64_t>(text_capacity, 0xffffff00ull);
    void *d_c = nullptr, *d_v = nullptr, *d_l = nullptr, *d_t = nullptr, *d_s = nullptr, *d_u = nullptr, *d_r = nullptr;
    auto done = [&](int rc) { for (void* q : {d_c, d_v, d_l, d_t, d_s, d_u, d_r}) if (q) (void)hipFree(q); return rc; };
    if (hipMalloc(&d_c, (size_t)n_lists * 4) != hipSuccess || hipMalloc(&d_v, slots * 8) != hipSuccess || hipMalloc(&d_l, slots * 8) != hipSuccess ||
        hipMalloc(&d_t, tcap + 16) != hipSuccess || hipMalloc(&d_s, (size_t)n_lists * 8) != hipSuccess || hipMalloc(&d_u, 64) != hipSuccess ||
        hipMalloc(&d_r, slots * 2 + 16) != hipSuccess)
        return done(fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc failed"));
    uint32_t cur[4] = {0, 0, 0, 0};
    bool ok = hipMemcpy(d_c, count, (size_t)n_lists * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(d_v, vaf, slots * 8, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_l, lnprob, slots * 8, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && vlr_launch_afd_text((const int32_t*)d_c, (const double*)d_v, (const double*)d_l, n_lists, capacity, (uint8_t*)d_t, (uint32_t)tcap, (uint32_t*)d_s, (uint16_t*)d_r, (uint32_t*)d_u, nullptr) == VLR_OK;
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(cur, d_u, 16, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(span, d_s, (size_t)n_lists * 8, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(text, d_t, std::min<size_t>(cur[0], tcap), hipMemcpyDeviceToHost) == hipSuccess;
    if (n_unformatted) *n_unformatted = cur[2];
    return done(ok ? VLR_OK : fail(VLR_ERR_HIP, "AFD text selftest: %s", hipGetErrorString(hipGetLastError())));
}

void* vlr_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(VLR_ERR_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void vlr_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// ---- read-vs-allele pair HMM (vlr_realign.hip)
extern "C" int vlr_launch_realign_kernel(const vlr_realign_batch_desc* b, double* ln_prob, void* stream);

static int check_realign(const vlr_realign_batch_desc* b, const double* ln_prob) {
    if (!b || (b->n_pairs > 0 && (!b->x_offset || !b->x_bases || !b->y_offset || !b->y_bases || !b->y_quals || !ln_prob)))
        return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (b->n_pairs < 0) return fail(VLR_ERR_INVALID_ARGUMENT, "negative pair count");
    for (int k = 0; k < 4; ++k)
        if (b->gap[k] != b->gap[k] || b->gap[k] > 0.0) return fail(VLR_ERR_INVALID_ARGUMENT, "gap[%d] is not a log probability", k);
    return VLR_OK;
}

extern "C" int vlr_launch_pathhmm_kernel(const vlr_realign_batch_desc* b, double* ln_prob, void* stream);
extern "C" int vlr_launch_homopoly_kernel(const vlr_realign_batch_desc* b, const double* hop, double* ln_prob, void* stream);
// the three modes behind one staging routine: `hop` is only read by the homopolymer launcher
struct realign_launcher {
    int (*plain)(const vlr_realign_batch_desc*, double*, void*);
    const double* hop;
    int operator()(const vlr_realign_batch_desc* b, double* out, void* st) const { return hop ? vlr_launch_homopoly_kernel(b, hop, out, st) : plain(b, out, st); }
};

static int realign_device(int device, const vlr_realign_batch_desc* b, double* ln_prob, void* hip_stream, realign_launcher launch) {
    int rc = check_realign(b, ln_prob);
    if (rc != VLR_OK) return rc;
    if (b->n_pairs == 0) return VLR_OK;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = (hipError_t)launch(b, ln_prob, hip_stream);
    if (e != hipSuccess) return fail(VLR_ERR_HIP, "realign kernel launch: %s", hipGetErrorString(e));
    return VLR_OK;
}
static int realign_host(int device, const vlr_realign_batch_desc* b, double* ln_prob, realign_launcher launch);

int vlr_realign_batch(int device, const vlr_realign_batch_desc* b, double* ln_prob, void* hip_stream) {
    return realign_device(device, b, ln_prob, hip_stream, realign_launcher{vlr_launch_realign_kernel, nullptr});
}
int vlr_realign_batch_host(int device, const vlr_realign_batch_desc* b, double* ln_prob) { return realign_host(device, b, ln_prob, realign_launcher{vlr_launch_realign_kernel, nullptr}); }
// `fast` realignment mode (PathHMMRealigner, realignment/mod.rs:547-678): best path probability over the alignments of minimal edit
// distance; max_edit_dist of the batch is not read
int vlr_realign_fast_batch(int device, const vlr_realign_batch_desc* b, double* ln_prob, void* hip_stream) {
    return realign_device(device, b, ln_prob, hip_stream, realign_launcher{vlr_launch_pathhmm_kernel, nullptr});
}
int vlr_realign_fast_batch_host(int device, const vlr_realign_batch_desc* b, double* ln_prob) { return realign_host(device, b, ln_prob, realign_launcher{vlr_launch_pathhmm_kernel, nullptr}); }
// `homopolymer` realignment mode (HomopolyPairHMMRealigner, realignment/mod.rs:680-730; HopParams pairhmm.rs:207-295)
static int check_hop(const double* hop) {
    if (!hop) return fail(VLR_ERR_INVALID_ARGUMENT, "null hop parameters");
    for (int k = 0; k < 16; ++k)
        if (!(hop[k] <= 0.0)) return fail(VLR_ERR_INVALID_ARGUMENT, "hop parameter %d is not a ln probability", k);
    return VLR_OK;
}
int vlr_realign_homopolymer_batch(int device, const vlr_realign_batch_desc* b, const double* hop, double* ln_prob, void* hip_stream) {
    int rc = check_hop(hop);
    if (rc != VLR_OK) return rc;
    return realign_device(device, b, ln_prob, hip_stream, realign_launcher{nullptr, hop});
}
int vlr_realign_homopolymer_batch_host(int device, const vlr_realign_batch_desc* b, const double* hop, double* ln_prob) {
    int rc = check_hop(hop);
    if (rc != VLR_OK) return rc;
    return realign_host(device, b, ln_prob, realign_launcher{nullptr, hop});
}

static int realign_host(int device, const vlr_realign_batch_desc* b, double* ln_prob, realign_launcher launch) {
    int rc = check_realign(b, ln_prob);
    if (rc != VLR_OK) return rc;
    const int64_t n = b->n_pairs;
    if (n == 0) return VLR_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return fail(VLR_ERR_NO_DEVICE, "no HIP device %d (the engine has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    const size_t nx = b->x_offset[n], ny = b->y_offset[n];
    // one staging buffer: offsets, bases, qualities, band, results
    const size_t o_xoff = 0, o_yoff = o_xoff + (n + 1) * 4, o_band = o_yoff + (n + 1) * 4, o_out = (o_band + n * 4 + 7) & ~(size_t)7;
    const size_t o_x = o_out + n * 8, o_y = o_x + ((nx + 7) & ~(size_t)7), o_q = o_y + ((ny + 7) & ~(size_t)7), total = o_q + ny + 8;
    char* d = nullptr;
    if (hipMalloc((void**)&d, total) != hipSuccess) { (void)hipGetLastError(); return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu)", total); }
    hipStream_t st = nullptr;
    rc = VLR_OK;
    do {
        if (hipMemcpyAsync(d + o_xoff, b->x_offset, (n + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_yoff, b->y_offset, (n + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_x, b->x_bases, nx, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_y, b->y_bases, ny, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_q, b->y_quals, ny, hipMemcpyHostToDevice, st) != hipSuccess ||
            (b->max_edit_dist && hipMemcpyAsync(d + o_band, b->max_edit_dist, n * 4, hipMemcpyHostToDevice, st) != hipSuccess)) {
            rc = fail(VLR_ERR_HIP, "staging copy failed"); break;
        }
        vlr_realign_batch_desc db = *b;
        db.x_offset = (const uint32_t*)(d + o_xoff); db.y_offset = (const uint32_t*)(d + o_yoff);
        db.x_bases = (const uint8_t*)(d + o_x); db.y_bases = (const uint8_t*)(d + o_y); db.y_quals = (const uint8_t*)(d + o_q);
        db.max_edit_dist = b->max_edit_dist ? (const int32_t*)(d + o_band) : nullptr;
        hipError_t e = (hipError_t)launch(&db, (double*)(d + o_out), st);
        if (e != hipSuccess) { rc = fail(VLR_ERR_HIP, "realign kernel launch: %s", hipGetErrorString(e)); break; }
        if (hipMemcpyAsync(ln_prob, d + o_out, n * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = fail(VLR_ERR_HIP, "result copy failed"); break;
        }
    } while (0);
    (void)hipFree(d);
    return rc;
}

// ---- edit-distance pre-filter (vlr_realign.hip)
extern "C" int vlr_launch_edit_kernel(const vlr_realign_batch_desc* b, int32_t* dist, int32_t* end, int32_t* n_hits, void* stream);

static int check_edit(const vlr_realign_batch_desc* b, const int32_t* dist) {
    if (!b || (b->n_pairs > 0 && (!b->x_offset || !b->x_bases || !b->y_offset || !b->y_bases || !dist)))
        return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    if (b->n_pairs < 0) return fail(VLR_ERR_INVALID_ARGUMENT, "negative pair count");
    return VLR_OK;
}

int vlr_edit_distance_batch(int device, const vlr_realign_batch_desc* b, int32_t* dist, int32_t* end, int32_t* n_hits, void* hip_stream) {
    int rc = check_edit(b, dist);
    if (rc != VLR_OK) return rc;
    if (b->n_pairs == 0) return VLR_OK;
    HIP_TRY(hipSetDevice(device));
    hipError_t e = (hipError_t)vlr_launch_edit_kernel(b, dist, end, n_hits, hip_stream);
    if (e != hipSuccess) return fail(VLR_ERR_HIP, "edit-distance kernel launch: %s", hipGetErrorString(e));
    return VLR_OK;
}

int vlr_edit_distance_batch_host(int device, const vlr_realign_batch_desc* b, int32_t* dist, int32_t* end, int32_t* n_hits) {
    int rc = check_edit(b, dist);
    if (rc != VLR_OK) return rc;
    const int64_t n = b->n_pairs;
    if (n == 0) return VLR_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return fail(VLR_ERR_NO_DEVICE, "no HIP device %d (the engine has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    const size_t nx = b->x_offset[n], ny = b->y_offset[n];
    const size_t o_xoff = 0, o_yoff = o_xoff + (n + 1) * 4, o_out = o_yoff + (n + 1) * 4;  // three int32 result arrays
    const size_t o_x = (o_out + 3 * n * 4 + 7) & ~(size_t)7, o_y = o_x + ((nx + 7) & ~(size_t)7), total = o_y + ny + 8;
    char* d = nullptr;
    if (hipMalloc((void**)&d, total) != hipSuccess) { (void)hipGetLastError(); return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu)", total); }
    hipStream_t st = nullptr;
    rc = VLR_OK;
    do {
        if (hipMemcpyAsync(d + o_xoff, b->x_offset, (n + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_yoff, b->y_offset, (n + 1) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_x, b->x_bases, nx, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d + o_y, b->y_bases, ny, hipMemcpyHostToDevice, st) != hipSuccess) {
            rc = fail(VLR_ERR_HIP, "staging copy failed"); break;
        }
        vlr_realign_batch_desc db = *b;
        db.x_offset = (const uint32_t*)(d + o_xoff); db.y_offset = (const uint32_t*)(d + o_yoff);
        db.x_bases = (const uint8_t*)(d + o_x); db.y_bases = (const uint8_t*)(d + o_y);
        int32_t* dd = (int32_t*)(d + o_out);
        hipError_t e = (hipError_t)vlr_launch_edit_kernel(&db, dd, dd + n, dd + 2 * n, st);
        if (e != hipSuccess) { rc = fail(VLR_ERR_HIP, "edit-distance kernel launch: %s", hipGetErrorString(e)); break; }
        if (hipMemcpyAsync(dist, dd, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            (end && hipMemcpyAsync(end, dd + n, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess) ||
            (n_hits && hipMemcpyAsync(n_hits, dd + 2 * n, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess) ||
            hipStreamSynchronize(st) != hipSuccess) {
            rc = fail(VLR_ERR_HIP, "result copy failed"); break;
        }
    } while (0);
    (void)hipFree(d);
    return rc;
}

// ---- Bayesian FDR threshold (vlr_fdr.hip)
extern "C" int vlr_launch_fdr(double* keys, long long n, long long npad, int smart, double alpha_ln, double* work, long long* best, double* fdr0, void* stream);

// The reference's accumulation (bio expected_fdr: ln_cumsum_exp = running ln_add_exp, minus ln rank, capped at ln 1) and its boundary
// search (fdr.rs:118-141) over the device-sorted arrays; only used when an expected FDR sits within rounding of alpha.
static void fdr_search_reference_order(const double* prob, int64_t n, double alpha_ln, double* threshold, int* status) {
    const double ninf = -std::numeric_limits<double>::infinity();
    // the PEPs with the host's libm as well (LogProb::ln_one_minus_exp): the device's log1p/exp may differ in the last bit
    std::vector<double> pep_ln((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const double q = prob[i];
        pep_ln[(size_t)i] = q >= 0.0 ? ninf : (q < -0.693 ? std::log1p(-std::exp(q)) : std::log(-std::expm1(q)));
    }
    double acc = ninf;
    int64_t best = -1;
    double f0 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double p = pep_ln[(size_t)i];
        const double hi = acc > p ? acc : p, lo = acc > p ? p : acc;
        acc = (hi == ninf) ? ninf : (lo == ninf ? hi : hi + std::log1p(std::exp(lo - hi)));   // LogProb::ln_add_exp
        double f = acc - std::log((double)(i + 1));
        f = f > 0.0 ? 0.0 : f;
        if (i == 0) f0 = f;
        if (f <= alpha_ln && (i == 0 || pep_ln[i] != pep_ln[i - 1])) best = i;
    }
    if (f0 > alpha_ln) { *status = VLR_FDR_LN_ONE; *threshold = 0.0; }
    else if (best < 0) { *status = VLR_FDR_NONE; *threshold = 0.0; }
    else { *status = VLR_FDR_VALUE; *threshold = prob[best]; }
}

int vlr_fdr_threshold(int device, const double* ln_prob, int64_t n, int smart, double alpha_ln, double* threshold, int* status) {
    if (n < 0 || (n > 0 && !ln_prob) || !threshold || !status) return fail(VLR_ERR_INVALID_ARGUMENT, "null argument");
    *threshold = 0.0;
    *status = VLR_FDR_EMPTY;
    if (n == 0) return VLR_OK;
    // LogProb sums of PHRED-rounded probabilities overshoot ln 1 by rounding: capped like cap_numerical_overshoot(NUMERICAL_EPSILON)
    // (utils/mod.rs:40, 207-209) instead of being refused; anything larger is not a log probability.
    std::vector<double> capped;
    for (int64_t i = 0; i < n; ++i) {
        if (ln_prob[i] != ln_prob[i] || ln_prob[i] > 1e-3) return fail(VLR_ERR_INVALID_ARGUMENT, "ln_prob[%lld] is not a log probability", (long long)i);
        if (ln_prob[i] > 0.0 && capped.empty()) capped.assign(ln_prob, ln_prob + n);
    }
    if (!capped.empty()) {
        for (auto& v : capped) v = v > 0.0 ? 0.0 : v;
        ln_prob = capped.data();
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return fail(VLR_ERR_NO_DEVICE, "no HIP device %d (the engine has no CPU path)", device);
    HIP_TRY(hipSetDevice(device));
    long long npad = 2048;
    while (npad < n) npad <<= 1;
    const long long nb = (n + 1023) / 1024;
    const size_t words = (size_t)npad + 3 * (size_t)n + (size_t)nb + 4;
    double* d = nullptr;
    if (hipMalloc((void**)&d, words * 8) != hipSuccess) { (void)hipGetLastError(); return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc(%zu)", words * 8); }
    int rc = VLR_OK;
    do {
        std::vector<double> pad((size_t)(npad - n), -std::numeric_limits<double>::infinity());
        double* keys = d;
        double* work = d + npad;
        long long* best = (long long*)(work + 3 * (size_t)n + nb);   // [0] boundary index + 1, [1] expected FDR of entry 0 (double), [2] near-alpha flag
        double* fdr0 = (double*)(best + 1);
        long long zero[3] = {0, 0, 0};
        if (hipMemcpy(keys, ln_prob, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
            (npad > n && hipMemcpy(keys + n, pad.data(), pad.size() * 8, hipMemcpyHostToDevice) != hipSuccess) ||
            hipMemcpy(best, zero, 24, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(VLR_ERR_HIP, "staging copy failed"); break; }
        hipError_t e = (hipError_t)vlr_launch_fdr(keys, (long long)n, npad, smart ? 1 : 0, alpha_ln, work, best, fdr0, nullptr);
        if (e != hipSuccess) { rc = fail(VLR_ERR_HIP, "fdr kernels: %s", hipGetErrorString(e)); break; }
        long long b[3] = {0, 0, 0};
        if (hipMemcpy(b, best, 24, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(VLR_ERR_HIP, "result copy failed"); break; }
        double f0;
        std::memcpy(&f0, &b[1], 8);
        if (b[2] != 0) {
            // an expected FDR within rounding of alpha: decide with the reference's accumulation order (ties at alpha)
            std::vector<double> host((size_t)n);
            if (hipMemcpy(host.data(), work, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(VLR_ERR_HIP, "result copy failed"); break; }
            fdr_search_reference_order(host.data(), n, alpha_ln, threshold, status);
            break;
        }
        if (f0 > alpha_ln) { *status = VLR_FDR_LN_ONE; *threshold = 0.0; }       // fdr.rs:127-128
        else if (b[0] == 0) { *status = VLR_FDR_NONE; }
        else {
            double pv = 0.0;
            if (hipMemcpy(&pv, work + (b[0] - 1), 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(VLR_ERR_HIP, "result copy failed"); break; }
            *status = VLR_FDR_VALUE; *threshold = pv;
        }
    } while (0);
    (void)hipFree(d);
    return rc;
}

// ---- diagnostics (tests/test_gpu_math.py)
extern "C" int vlr_launch_selftest_math(int which, const double* a, const double* b, double* out, long long n, void* stream);
int vlr_selftest_math(int device, int which, const double* a, const double* b, double* out, int64_t n) {
    if (n < 0 || (n > 0 && (!a || !out)) || which < 0 || which > 4) return fail(VLR_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return VLR_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return fail(VLR_ERR_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)n * 24));
    int rc = VLR_OK;
    if (hipMemcpy(d, a, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, b ? b : a, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(VLR_ERR_HIP, "copy failed");
    if (rc == VLR_OK && vlr_launch_selftest_math(which, d, d + n, d + 2 * n, (long long)n, nullptr) != 0) rc = fail(VLR_ERR_HIP, "launch failed");
    if (rc == VLR_OK && hipMemcpy(out, d + 2 * n, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(VLR_ERR_HIP, "copy back failed");
    (void)hipFree(d);
    return rc;
}

extern "C" int vlr_launch_selftest_stream(const float* in, double* out, long long n, int mode, void* stream);
int vlr_selftest_stream(int device, int mode, int64_t n, int reps) {
    if (n <= 0 || reps <= 0 || (mode != 0 && mode != 1)) return fail(VLR_ERR_INVALID_ARGUMENT, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return fail(VLR_ERR_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    float* in = nullptr;
    double* out = nullptr;
    HIP_TRY(hipMalloc((void**)&in, (size_t)n * 4));
    if (hipMalloc((void**)&out, mode == 1 ? (size_t)n * 8 : (size_t)(n / 16384 + 1) * 8) != hipSuccess) { (void)hipFree(in); return fail(VLR_ERR_OUT_OF_MEMORY, "hipMalloc"); }
    (void)hipMemset(in, 0, (size_t)n * 4);
    int rc = VLR_OK;
    for (int r = 0; r < reps && rc == VLR_OK; ++r)
        if (vlr_launch_selftest_stream(in, out, (long long)n, mode, nullptr) != 0) rc = fail(VLR_ERR_HIP, "launch failed");
    if (hipDeviceSynchronize() != hipSuccess && rc == VLR_OK) rc = fail(VLR_ERR_HIP, "kernel failed");
    (void)hipFree(in); (void)hipFree(out);
    return rc;
}

}  // extern "C"
