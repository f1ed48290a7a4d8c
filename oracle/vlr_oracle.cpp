// vlr_oracle.cpp — CPU restatement (f64, single-threaded per call) of the per-locus model
// evaluation of `varlociraptor call variants` (reference v8.9.3, /root/reference).
//
// THIS IS TEST INFRASTRUCTURE.  It is the parity oracle for the HIP path and the CPU baseline of
// bench.py (`cpu_baseline.kind = "port"`).  Nothing in the product (varlociraptor_amd/, the C-ABI
// library) may include, link or call it.
//
// Parity status: PINNED for the clean single-sample range path by the reference's only numeric
// known-answer fixture tests/resources/flamegraph_profiling/{normal.vcf -> calls.vcf}
// (tests/test_oracle_fixture.py); bias mixtures, contamination, multi-sample trees, Set spectra,
// LFC pruning and the Mendelian prior have NO numeric golden in the reference ("parity unpinned" for
// those; fidelity argued by line-by-line correspondence + the unit identities of likelihood.rs:273-394).
//
// Third-party arithmetic not present under /root/reference (Cargo.toml:29 bio = "2.0.0";
// itertools-num 0.1; statrs 0.18; approx) is restated from the published crate behaviour:
//   bio::stats::LogProb::{ln_one_minus_exp, ln_add_exp, ln_sum_exp, ln_simpsons_integrate_exp,
//   ln_trapezoidal_integrate_grid_exp}, bio::stats::bayesian::model::Model::compute,
//   BayesFactor / KassRaftery, itertools_num::linspace, statrs Hypergeometric::pmf,
//   approx::relative_eq! defaults.
//
// Deliberate deviations from reference *quirks* (all documented in DESIGN.md):
//   * HashMap iteration orders (event order, argmax ties in adaptive integration, MAP ties) are
//     replaced by deterministic rules: events in given order; lowest index wins the bracket argmax;
//     MAP ties broken by (non-artifact first, then smaller VAF tuple).
//   * the prior LRU cache is not shared across variant types (calling.rs:414-426 shares a Model and its
//     prior cache between SNV and MNV records; results there depend on record order).
//
// Every function cites the reference file:line it follows.

#include "../include/vlr.h"
#include "../include/vlr_detmath.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

const double NEG_INF = -std::numeric_limits<double>::infinity();
const double LN_05 = std::log(0.5);    // utils/mod.rs:45 PROB_05 = LogProb::from(Prob(0.5))
const double LN_095 = std::log(0.95);  // utils/mod.rs:48 PROB_095

// ------------------------------------------------------------------ bio::stats::LogProb (restated)
inline double ln_one_minus_exp(double p) {
    // bio LogProb::ln_one_minus_exp -> ln_1m_exp
    if (p < -0.693) return std::log1p(-std::exp(p));
    return std::log(-std::expm1(p));
}
inline double ln_add_exp(double a, double b) {
    double p0 = a, p1 = b;
    if (p1 > p0) std::swap(p0, p1);
    if (p0 == NEG_INF) return NEG_INF;
    if (p0 == std::numeric_limits<double>::infinity()) return p0;
    return p0 + std::log1p(std::exp(p1 - p0));
}
inline double ln_sum_exp(const std::vector<double>& v) {
    if (v.empty()) return NEG_INF;
    double pmax = v[0];
    size_t imax = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > pmax) { pmax = v[i]; imax = i; }
    if (pmax == NEG_INF) return NEG_INF;
    if (pmax == std::numeric_limits<double>::infinity()) return pmax;
    double s = 0.0;
    for (size_t i = 0; i < v.size(); ++i) {
        if (i == imax || v[i] == NEG_INF) continue;
        s += std::exp(v[i] - pmax);
    }
    return pmax + std::log1p(s);
}
// approx::relative_eq!(a, b) with default epsilon = max_relative = f64::EPSILON
inline bool relative_eq(double a, double b) {
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;
    double d = std::fabs(a - b);
    const double eps = std::numeric_limits<double>::epsilon();
    if (d <= eps) return true;
    double largest = std::max(std::fabs(a), std::fabs(b));
    return d <= largest * eps;
}
// itertools_num::linspace(a, b, n)
inline std::vector<double> linspace(double a, double b, int n) {
    std::vector<double> out(n);
    double step = (n > 1) ? (b - a) / (double)(n - 1) : 0.0;
    for (int i = 0; i < n; ++i) out[i] = a + step * (double)i;
    return out;
}

// ------------------------------------------------------------------ observations
struct Obs {  // read_observation.rs:219-278 (ProcessedReadObservation)
    double prob_mapping, prob_mismapping;
    double prob_alt, prob_ref;               // originals
    bool has_adj = false;                    // prob_alt_adj / prob_ref_adj (402-415)
    double prob_alt_adj = 0, prob_ref_adj = 0;
    double prob_missed_allele, prob_sample_alt, prob_double_overlap, prob_single_overlap, prob_hit_base;
    int strand, orientation;
    bool readpos_major, softclipped, paired, is_max_mapq;
    int alt_locus;
    bool has_hp_art, has_hp_var, has_hp_len;
    double hp_art, hp_var;
    int hp_len;

    double p_alt() const { return has_adj ? prob_alt_adj : prob_alt; }  // read_observation.rs:409-411
    double p_ref() const { return has_adj ? prob_ref_adj : prob_ref; }  // 413-415
    // BayesFactor::new(a,b) = exp(a-b); KassRaftery: >20 Strong, >3 Positive (bio)
    bool is_uniquely_mapping() const { return prob_mapping >= LN_095; }                    // 425-427
    bool is_strong_alt_support() const { return std::exp(prob_alt - prob_ref) > 20.0; }   // 429-432
    bool is_strong_ref_support() const { return std::exp(prob_ref - prob_alt) > 20.0; }   // 434-437
    bool is_ref_support() const { return prob_ref > prob_alt; }                            // 439-441
    bool is_positive_ref_support() const { return std::exp(prob_ref - prob_alt) > 3.0; }  // 443-446
};
struct Pileup {
    std::vector<Obs> obs;
    size_t n_filtered = 0;
};

// ------------------------------------------------------------------ biases (variants/model/bias/*.rs)
struct Artifacts {  // bias/mod.rs:114-129
    int sb = 0;     // 0 None{forward_rate} 1 Forward 2 Reverse   (strand_bias.rs:14-19)
    double forward_rate = 0.5;
    int rob = 0;    // 0 None 1 F1R2 2 F2R1                        (read_orientation_bias.rs:9-15)
    int rpb = 0;    // 0 None 1 Some                               (read_position_bias.rs:10-15)
    int scb = 0;    //                                               (softclip_bias.rs:7-12)
    int he = 0;     //                                               (homopolymer_error.rs:9-14)
    int alb = 0;    // 0 None 1 Some{has_alt_loci}                 (alt_locus_bias.rs:10-17)
    bool has_alt_loci = false;
    int id = 0;     // index within the learned hypothesis list (0 = none); used for cache keys

    bool is_artifact() const { return sb || rob || rpb || scb || he || alb; }  // bias/mod.rs:286-293
};

// strand_bias.rs:30-54
double sb_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.sb == 1) {  // Forward
        if (o.strand == VLR_STRAND_FORWARD) return 0.0;
        if (o.strand == VLR_STRAND_REVERSE) return NEG_INF;
        if (o.strand == VLR_STRAND_BOTH) return NEG_INF;
        return 0.0;  // (_, Strand::None)
    }
    if (a.sb == 2) {  // Reverse
        if (o.strand == VLR_STRAND_FORWARD) return NEG_INF;
        if (o.strand == VLR_STRAND_REVERSE) return 0.0;
        if (o.strand == VLR_STRAND_BOTH) return NEG_INF;
        return 0.0;
    }
    if (o.strand == VLR_STRAND_BOTH) return o.prob_double_overlap;
    if (o.strand == VLR_STRAND_NONE) return 0.0;
    double rate = (o.strand == VLR_STRAND_FORWARD) ? a.forward_rate : 1.0 - a.forward_rate;
    return std::log(rate) + o.prob_single_overlap;
}
// read_orientation_bias.rs:18-32
double rob_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.rob == 0) return LN_05;
    if (a.rob == 1) {
        if (o.orientation == VLR_ORIENT_F1R2) return 0.0;
        if (o.orientation == VLR_ORIENT_F2R1) return NEG_INF;
        return LN_05;
    }
    if (o.orientation == VLR_ORIENT_F2R1) return 0.0;
    if (o.orientation == VLR_ORIENT_F1R2) return NEG_INF;
    return LN_05;
}
// read_position_bias.rs:51-62
double one_minus_prob_hit_base(const Obs& o) {
    if (o.prob_hit_base != 0.0) return ln_one_minus_exp(o.prob_hit_base);
    return 0.0;
}
// read_position_bias.rs:18-37
double rpb_prob_any(const Obs& o) { return o.readpos_major ? o.prob_hit_base : one_minus_prob_hit_base(o); }
double rpb_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.rpb == 0) return rpb_prob_any(o);
    return o.readpos_major ? 0.0 : NEG_INF;
}
// softclip_bias.rs:15-25
double scb_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.scb == 1) return o.softclipped ? 0.0 : NEG_INF;
    return 0.0;
}
// homopolymer_error.rs:23-40
double he_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.he == 1) return o.has_hp_art ? o.hp_art : 0.0;
    return o.has_hp_var ? o.hp_var : 0.0;
}
// alt_locus_bias.rs:63-109
double alb_prob_alt(const Artifacts& a, const Obs& o) {
    if (a.alb == 1) {
        if (a.has_alt_loci) return (o.alt_locus == VLR_ALTLOCUS_MAJOR) ? 0.0 : NEG_INF;
        return o.is_max_mapq ? NEG_INF : 0.0;
    }
    return LN_05;
}
double alb_prob_ref(const Artifacts& a, const Obs& o) {
    if (a.alb == 1) {
        if (a.has_alt_loci) return (o.alt_locus == VLR_ALTLOCUS_MAJOR) ? NEG_INF : 0.0;
        return LN_05;
    }
    return LN_05;
}
// bias/mod.rs:259-284 (sum order as in the reference)
double art_prob_alt(const Artifacts& a, const Obs& o) {
    return sb_prob_alt(a, o) + rob_prob_alt(a, o) + rpb_prob_alt(a, o) + scb_prob_alt(a, o) + he_prob_alt(a, o) +
           alb_prob_alt(a, o);
}
double art_prob_ref(const Artifacts& a, const Obs& o) {
    // Bias::prob_ref defaults to prob_any (bias/mod.rs:29-31); homopolymer: prob_ref = prob_alt
    // (homopolymer_error.rs:34-36); alt locus: own prob_ref (alt_locus_bias.rs:86-109)
    return LN_05 + LN_05 + rpb_prob_any(o) + 0.0 + he_prob_alt(a, o) + alb_prob_ref(a, o);
}
double art_prob_any(const Artifacts&, const Obs& o) {
    // strand .5 (strand_bias.rs:56-58), orientation .5, position (read_position_bias.rs:27-37),
    // softclip 1, homopolymer 1, alt locus .5
    return LN_05 + LN_05 + rpb_prob_any(o) + 0.0 + 0.0 + LN_05;
}

// single artifact component selector for the gating logic: which of the six is the artifact one
enum Comp { C_SB = 0, C_ROB, C_RPB, C_SCB, C_HE, C_ALB };
double comp_prob_alt(int comp, const Artifacts& a, const Obs& o) {
    switch (comp) {
        case C_SB: return sb_prob_alt(a, o);
        case C_ROB: return rob_prob_alt(a, o);
        case C_RPB: return rpb_prob_alt(a, o);
        case C_SCB: return scb_prob_alt(a, o);
        case C_HE: return he_prob_alt(a, o);
        default: return alb_prob_alt(a, o);
    }
}
bool comp_is_artifact(int comp, const Artifacts& a) {
    switch (comp) {
        case C_SB: return a.sb != 0;
        case C_ROB: return a.rob != 0;
        case C_RPB: return a.rpb != 0;
        case C_SCB: return a.scb != 0;
        case C_HE: return a.he != 0;
        default: return a.alb != 0;
    }
}

// exp(LogProb::ln_sum_exp(v)) as used by strand_bias.rs:80-109 and read_position_bias.rs:68-113, evaluated with the
// platform-independent exp/log1p of include/vlr_detmath.h (same formula m + ln1p(sum_{i != imax} exp(v_i - m)));
// see that header for why these decision sums must not depend on libm's last bit.
double exp_lse_det(const std::vector<double>& v) {
    if (v.empty()) return 0.0;
    double m = NEG_INF;
    for (double x : v) m = std::max(m, x);
    if (m == NEG_INF) return 0.0;
    vlr_det::dd s{0.0, 0.0};  // order-independent (double-double) sum, see include/vlr_detmath.h
    for (double x : v)
        if (x != NEG_INF) s = vlr_det::dd_add(s, vlr_det::det_exp(x - m));
    return vlr_det::exp_lse_from_sum(m, s);
}

// strand_bias.rs:79-123
bool estimate_forward_rate(const std::vector<Pileup>& pileups, double* rate) {
    std::vector<double> all, fwd;
    for (auto& p : pileups)
        for (auto& o : p.obs) {
            if (o.is_strong_ref_support() && o.strand != VLR_STRAND_BOTH) all.push_back(o.prob_mapping);
            if (o.is_strong_ref_support() && o.strand == VLR_STRAND_FORWARD) fwd.push_back(o.prob_mapping);
        }
    double strong_all = exp_lse_det(all);
    double strong_forward = exp_lse_det(fwd);
    if (strong_all > 2.0) {
        double f = strong_forward / strong_all;
        if (strong_all > 100.0 && f > 0.0 && f < 1.0) { *rate = f; return true; }
        if (f >= 0.4 && f <= 0.6) { *rate = 0.5; return true; }
    }
    return false;
}
// alt_locus_bias.rs:47-60
bool has_alt_loci(const std::vector<Pileup>& pileups) {
    for (auto& p : pileups)
        for (auto& o : p.obs)
            if (o.alt_locus != VLR_ALTLOCUS_NONE) return true;
    return false;
}
// read_position_bias.rs:63-122
bool has_valid_major_rate(const std::vector<Pileup>& pileups) {
    for (auto& p : pileups) {
        std::vector<double> all, major, rate;
        for (auto& o : p.obs)
            if (o.is_strong_ref_support()) {
                all.push_back(o.prob_mapping);
                if (o.readpos_major) major.push_back(o.prob_mapping);
                rate.push_back(o.prob_mapping + o.prob_hit_base);
            }
        double expected_all = exp_lse_det(all);
        if (expected_all > 10.0) {
            double expected_major = exp_lse_det(major);
            double expected_major_rate = exp_lse_det(rate);
            double major_rate = expected_major / expected_all;
            if (expected_major > 0.0 && std::fabs(major_rate - expected_major_rate) < 0.05) return true;
        }
    }
    return false;
}
// homopolymer_error.rs:46-72
bool he_is_informative(const std::vector<Pileup>& pileups) {
    for (auto& p : pileups) {
        bool any_strong_alt = false, ins = false, del = false;
        for (auto& o : p.obs) {
            if (o.is_strong_alt_support()) any_strong_alt = true;
            int l = o.has_hp_len ? o.hp_len : 0;
            if (l > 0) ins = true;
            if (l < 0) del = true;
        }
        if (!(!any_strong_alt || (ins && del))) return false;
    }
    return true;
}
// read_orientation_bias.rs:38-97
bool rob_is_informative(const std::vector<Pileup>& pileups) {
    size_t n_uncertain = 0, n = 0, strong_ref_total = 0, strong_ref_f1r2 = 0;
    for (auto& p : pileups) {
        n += p.obs.size();
        for (auto& o : p.obs) {
            bool std_or = (o.orientation == VLR_ORIENT_F1R2 || o.orientation == VLR_ORIENT_F2R1);
            if (!std_or) n_uncertain++;
            if (o.is_strong_ref_support() && std_or) strong_ref_total++;
            if (o.is_strong_ref_support() && o.orientation == VLR_ORIENT_F1R2) strong_ref_f1r2++;
        }
    }
    bool enough_information = (double)n_uncertain < ((double)n / 2.0);
    bool uniform = false;
    if (strong_ref_total > 2) {
        double fraction = (double)strong_ref_f1r2 / (double)strong_ref_total;
        uniform = fraction >= 0.3 && fraction <= 0.7;
    }
    return enough_information && uniform;
}
// alt_locus_bias.rs:124-144
bool alb_is_informative(const std::vector<Pileup>& pileups) {
    size_t n_alt = 0, nm_alt = 0, n_ref = 0, nm_ref = 0;
    for (auto& p : pileups)
        for (auto& o : p.obs) {
            if (o.is_strong_alt_support()) { n_alt++; if (!o.is_max_mapq) nm_alt++; }
            if (o.is_strong_ref_support()) { n_ref++; if (!o.is_max_mapq) nm_ref++; }
        }
    bool enough_alt = n_alt > 0 && (double)nm_alt > ((double)n_alt * 0.1) && (n_alt - nm_alt) < 10;
    bool enough_ref = n_ref > 0 && ((double)nm_ref < ((double)n_ref * 0.9));
    return enough_alt && (has_alt_loci(pileups) || enough_ref);
}

// Bias::is_possible default (bias/mod.rs:37-48) for the artifact component
bool comp_is_possible(int comp, const Artifacts& a, const std::vector<Pileup>& pileups) {
    if (!comp_is_artifact(comp, a)) return true;
    if (comp == C_HE) return he_is_informative(pileups);  // homopolymer_error.rs:74-76
    for (auto& p : pileups)
        for (auto& o : p.obs)
            if (comp_prob_alt(comp, a, o) != NEG_INF) return true;
    return false;
}
bool comp_is_informative(int comp, const Artifacts& a, const std::vector<Pileup>& pileups) {
    if (!comp_is_artifact(comp, a)) return true;
    switch (comp) {
        case C_SB: { double r; return estimate_forward_rate(pileups, &r); }  // strand_bias.rs:59-64
        case C_ROB: return rob_is_informative(pileups);
        case C_RPB: return has_valid_major_rate(pileups);                    // read_position_bias.rs:43-47
        case C_SCB: {                                                        // softclip_bias.rs:31-39
            for (auto& p : pileups)
                for (auto& o : p.obs)
                    if (o.softclipped) return true;
            return false;
        }
        case C_HE: return he_is_informative(pileups);
        default: return alb_is_informative(pileups);
    }
}
// Bias::is_likely default (bias/mod.rs:60-104)
bool comp_is_likely(int comp, const Artifacts& a, const std::vector<Pileup>& pileups) {
    if (!comp_is_artifact(comp, a)) return true;
    if (comp == C_HE) return he_is_informative(pileups);  // homopolymer_error.rs:78-80
    double min_ratio = 0.66666;                           // bias/mod.rs:56-58
    for (auto& p : pileups) {
        size_t strong_all = 0, strong_bias = 0;
        bool all_ref = true;
        for (auto& o : p.obs) {
            bool s = o.is_uniquely_mapping() && o.is_strong_alt_support();
            if (s) {
                strong_all++;
                if (comp_prob_alt(comp, a, o) != NEG_INF) strong_bias++;  // is_bias_evidence (bias/mod.rs:52-54)
            }
            if (!o.is_ref_support()) all_ref = false;
        }
        bool r;
        if (strong_all >= 10) {
            double ratio = (double)strong_bias / (double)strong_all;
            r = ratio >= min_ratio;
        } else if (all_ref) {
            r = false;
        } else if (p.obs.empty()) {
            r = false;
        } else {
            r = true;
        }
        if (r) return true;
    }
    return false;
}
// bias/mod.rs:232-257
bool art_gate(const Artifacts& a, const std::vector<Pileup>& pileups) {
    for (int c = 0; c < 6; ++c)
        if (!comp_is_possible(c, a, pileups)) return false;
    for (int c = 0; c < 6; ++c)
        if (!comp_is_informative(c, a, pileups)) return false;
    for (int c = 0; c < 6; ++c)
        if (!comp_is_likely(c, a, pileups)) return false;
    return true;
}
// bias/mod.rs:295-300 + strand_bias.rs:66-76 + alt_locus_bias.rs:115-122
void art_learn(Artifacts& a, const std::vector<Pileup>& pileups) {
    if (a.sb == 0) {
        double r;
        a.forward_rate = estimate_forward_rate(pileups, &r) ? r : 0.5;
    }
    if (a.alb == 1) a.has_alt_loci = has_alt_loci(pileups);
}
// bias/mod.rs:131-218: all combinations with exactly one artifact component, in cartesian order
// (strand outermost ... alt-locus innermost)
std::vector<Artifacts> all_artifact_combinations(unsigned mask) {
    std::vector<Artifacts> out;
    if (!(mask & 0x3f)) return out;
    int nsb = (mask & VLR_BIAS_STRAND) ? 3 : 1;
    int nrob = (mask & VLR_BIAS_ORIENTATION) ? 3 : 1;
    int nrpb = (mask & VLR_BIAS_POSITION) ? 2 : 1;
    int nscb = (mask & VLR_BIAS_SOFTCLIP) ? 2 : 1;
    int nhe = (mask & VLR_BIAS_HOMOPOLYMER) ? 2 : 1;
    int nalb = (mask & VLR_BIAS_ALTLOCUS) ? 2 : 1;
    for (int sb = 0; sb < nsb; ++sb)
        for (int rob = 0; rob < nrob; ++rob)
            for (int rpb = 0; rpb < nrpb; ++rpb)
                for (int scb = 0; scb < nscb; ++scb)
                    for (int he = 0; he < nhe; ++he)
                        for (int alb = 0; alb < nalb; ++alb) {
                            int n = (sb != 0) + (rob != 0) + (rpb != 0) + (scb != 0) + (he != 0) + (alb != 0);
                            if (n != 1) continue;
                            Artifacts a;
                            a.sb = sb; a.rob = rob; a.rpb = rpb; a.scb = scb; a.he = he; a.alb = alb;
                            out.push_back(a);
                        }
    return out;
}

// ------------------------------------------------------------------ spectra (grammar/formula.rs:1018-1262)
struct Range {
    double start, end;
    bool lex, rex;
    static Range empty() { return {0.0, 0.0, true, true}; }                                 // 1070-1076
    bool is_empty() const { return start == end && (lex || rex); }                           // 1078-1080
    bool is_singleton() const { return start == end && !(lex || rex); }                      // 1086-1088
    bool contains(double v) const {                                                          // 1090-1097
        if (lex && rex) return start < v && end > v;
        if (lex && !rex) return start < v && end >= v;
        if (!lex && rex) return start <= v && end > v;
        return start <= v && end >= v;
    }
    bool operator==(const Range& o) const { return start == o.start && end == o.end && lex == o.lex && rex == o.rex; }
    // 1131-1168; returns true if overlap is None
    bool overlap_none(const Range& o) const {
        if (*this == o) return false;
        return (end < o.start || start > o.end) || (end <= o.start && (rex || o.lex)) ||
               (start >= o.end && (lex || o.rex));
    }
    static bool is_adjustment_possible(double start, double end, size_t n) { return (double)n * (end - start) > 1.0; }  // 1222-1224
    double observable_max(size_t n) const {                                                  // 1198-1216
        assert(end != 0.0);
        if (n < 10 || !is_adjustment_possible(start, end, n)) return end;
        double c = (double)n * end;
        if (rex && std::fmod(c, 1.0) == 0.0) c -= 1.0;
        c = std::floor(c);
        if (c == 0.0) return end;
        return std::floor(c) / (double)n;
    }
    double observable_min(size_t n) const {                                                  // 1170-1196
        double min_vaf;
        if (n < 10 || !is_adjustment_possible(start, end, n)) {
            min_vaf = start;
        } else {
            double c = (double)n * start;
            auto adjust = [&](double cc) { return std::ceil(cc) / (double)n; };
            if (lex && std::fmod(c, 1.0) == 0.0) {
                double adjusted_end = observable_max(n);
                for (double offset : {1.0, 0.0}) {
                    double s = adjust(c + offset);
                    if (s <= 1.0 && s <= adjusted_end) return s;
                }
            }
            min_vaf = adjust(c);
        }
        if (min_vaf >= observable_max(n)) return start;
        return min_vaf;
    }
    Range intersect(const Range& o) const {                                                  // 1226-1254
        if (overlap_none(o)) return empty();
        Range r;
        r.start = std::max(start, o.start);
        r.end = std::min(end, o.end);
        r.lex = (start > o.start) ? lex : (start < o.start) ? o.lex : (lex || o.lex);
        r.rex = (end < o.end) ? rex : (end > o.end) ? o.rex : (rex || o.rex);
        return r;
    }
};
struct Spectrum {
    bool is_set;
    std::vector<double> set;  // ascending (BTreeSet)
    Range range;
    bool contains(double v) const {  // 1035-1040
        if (is_set) return std::find(set.begin(), set.end(), v) != set.end();
        return range.contains(v);
    }
};
Spectrum spectrum_from(const vlr_spectrum& s, const double* pool) {
    Spectrum out;
    out.is_set = (s.kind == VLR_SPECTRUM_SET);
    if (out.is_set) {
        out.set.assign(pool + s.set_offset, pool + s.set_offset + s.set_len);
        out.range = Range::empty();
    } else {
        out.range = {s.start, s.end, s.left_exclusive != 0, s.right_exclusive != 0};
    }
    return out;
}

// ------------------------------------------------------------------ LFC (utils/log2_fold_change.rs)
struct LfcPred {
    int cmp;
    double value;
    bool operator==(const LfcPred& o) const { return cmp == o.cmp && value == o.value; }
    bool is_true(double vaf_a, double vaf_b) const {  // 17-26, 41-52
        double lfc = vlr_det::det_log2_ratio(vaf_a, vaf_b);  // see include/vlr_detmath.h: platform-independent at exact 2^k ratios
        switch (cmp) {
            case VLR_CMP_EQUAL: return relative_eq(lfc, value);
            case VLR_CMP_GREATER: return lfc > value;
            case VLR_CMP_GREATER_EQUAL: return lfc >= value;
            case VLR_CMP_LESS: return lfc < value;
            case VLR_CMP_LESS_EQUAL: return lfc <= value;
            default: return !relative_eq(lfc, value);
        }
    }
    Range infer_vaf_bounds(double vaf) const {  // 56-93
        double proj = vaf / vlr_det::det_exp2(value);  // platform-independent, exact for integral values
        if (proj < 0.0 || proj > 1.0) return Range::empty();
        switch (cmp) {
            case VLR_CMP_EQUAL: return {proj, proj, false, false};
            case VLR_CMP_GREATER: return {0.0, proj, false, true};
            case VLR_CMP_GREATER_EQUAL: return {0.0, proj, false, false};
            case VLR_CMP_LESS: return {proj, 1.0, true, false};
            case VLR_CMP_LESS_EQUAL: return {proj, 1.0, false, false};
            default: return {0.0, 1.0, false, false};
        }
    }
    LfcPred invert() const {  // 95-122
        switch (cmp) {
            case VLR_CMP_EQUAL: return {VLR_CMP_EQUAL, value};
            case VLR_CMP_GREATER: return {VLR_CMP_LESS_EQUAL, -value};
            case VLR_CMP_GREATER_EQUAL: return {VLR_CMP_LESS, -value};
            case VLR_CMP_LESS: return {VLR_CMP_GREATER_EQUAL, -value};
            case VLR_CMP_LESS_EQUAL: return {VLR_CMP_GREATER, -value};
            default: return {VLR_CMP_NOT_EQUAL, value};
        }
    }
};
struct VafLfc {  // modes/generic.rs:108-114
    int sample_a, sample_b;
    LfcPred pred;
    bool operator==(const VafLfc& o) const { return sample_a == o.sample_a && sample_b == o.sample_b && pred == o.pred; }
};

// ------------------------------------------------------------------ operands (modes/generic.rs:116-183)
struct SampleEvent {  // likelihood.rs:18-23
    bool present = false;
    double af = 0.0;
    int art = 0;  // index into Ctx::hyps (learned Artifacts)
    bool discrete = false;
};
struct Operands {
    std::vector<SampleEvent> events;  // VecMap keyed by sample
    std::vector<VafLfc> lfcs;
    size_t len() const {
        size_t n = 0;
        for (auto& e : events) n += e.present;
        return n;
    }
};

// ------------------------------------------------------------------ VAF tree (grammar/vaftree.rs)
struct Node {
    int kind;
    int sample = 0, sample_b = 0;
    LfcPred pred{0, 0.0};
    Spectrum vafs;
    bool positive = false;
    uint8_t refbase = 0, altbase = 0;
    std::vector<int> children;
};
struct Event {  // variants/model/mod.rs:35-39
    std::string name;
    std::vector<int> roots;
    std::vector<int> biases;  // hypothesis ids
    bool artifact;
    int named_index;          // -1 absent, else scenario event index
};

bool iupac_contains(uint8_t code, uint8_t base) {  // grammar/formula.rs:23-43
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

// ------------------------------------------------------------------ prior (variants/model/prior.rs)
struct Prior {
    int S = 0;
    std::vector<uint8_t> uniform;
    std::vector<int> ploidy;  // -1 none
    std::vector<std::vector<Spectrum>> universe;
    std::vector<double> germline_rate, somatic_rate;  // NaN none
    std::vector<vlr_inheritance> inheritance;
    double heterozygosity_ln = NAN;  // LogProb; NaN none
    double variant_heterozygosity_ln = NAN, variant_somatic_rate_ln = NAN;  // per-variant overrides (calling.rs:470-494)
    double frac_indel = 0.0125, frac_mnv = 0.001, frac_sv = 0.01;
    bool is_absent_only = true;
    int variant_type = VLR_VT_SNV;

    bool is_all_uniform() const {  // 111-113
        for (auto u : uniform)
            if (!u) return false;
        return true;
    }
    double variant_type_fraction() const {  // grammar/mod.rs:420-431
        switch (variant_type) {
            case VLR_VT_INDEL: return frac_indel;
            case VLR_VT_MNV: return frac_mnv;
            case VLR_VT_SV: return frac_sv;
            default: return 1.0;
        }
    }
    bool somatic_rate_ln(int s, double* out) const {  // 250-257
        if (!std::isnan(variant_somatic_rate_ln)) { *out = variant_somatic_rate_ln; return true; }
        if (std::isnan(somatic_rate[s])) return false;
        *out = std::log(somatic_rate[s] * variant_type_fraction());
        return true;
    }
    bool heterozygosity(double* out) const {  // 263-270
        if (!std::isnan(variant_heterozygosity_ln)) { *out = variant_heterozygosity_ln; return true; }
        if (std::isnan(heterozygosity_ln)) return false;
        *out = std::log(std::exp(heterozygosity_ln) * variant_type_fraction());
        return true;
    }
    bool is_valid_germline_vaf(int s, double vaf) const {  // 236-240
        double n_alt = (double)ploidy[s] * vaf;
        return relative_eq(n_alt, std::round(n_alt));
    }
    static double prob_somatic_mutation(double rate_ln, double somatic_vaf) {  // 440-456
        if (relative_eq(somatic_vaf, 0.0)) return ln_one_minus_exp(rate_ln);
        return rate_ln;
    }
    double eff_somatic(int s, const std::vector<double>& ev, const std::vector<double>& g) const { return ev[s] - g[s]; }  // 288-296
    double prob_clonal(int s, int parent, const std::vector<double>& ev, const std::vector<double>& g, bool somatic) const {  // 458-512
        if (!relative_eq(g[s], g[parent])) return NEG_INF;
        double r;
        bool has = somatic_rate_ln(s, &r);
        if (somatic && has) {
            double pv = eff_somatic(parent, ev, g), sv = eff_somatic(s, ev, g);
            if (pv != 0.0) return 0.0;
            return prob_somatic_mutation(r, sv);
        } else if (somatic && !has) {
            return relative_eq(eff_somatic(s, ev, g), eff_somatic(parent, ev, g)) ? 0.0 : NEG_INF;
        } else if (!somatic && has) {
            return prob_somatic_mutation(r, eff_somatic(s, ev, g));
        }
        return 0.0;
    }
    double prob_subclonal(int s, int parent, const std::vector<double>& ev, const std::vector<double>& g) const {  // 514-552
        double total = ev[s], germ = g[s];
        if (!relative_eq(germ, g[parent])) return NEG_INF;
        double parent_total = ev[parent];
        double r;
        if (somatic_rate_ln(s, &r)) {
            if (parent_total == 0.0 && germ == 0.0) return prob_somatic_mutation(r, total);
            return 0.0;
        }
        return relative_eq(eff_somatic(s, ev, g), eff_somatic(parent, ev, g)) ? 0.0 : NEG_INF;
    }
    double prob_population_germline(const std::vector<int>& pop, const std::vector<double>& g, double het) const {  // 554-582
        unsigned m = 0;
        for (int s : pop) m += (unsigned)std::llround((double)ploidy[s] * g[s]);
        auto prob_m = [&](unsigned mm) { return het - std::log((double)mm); };
        if (m > 0) return prob_m(m);
        unsigned n = 0;
        for (int s : pop) n += (unsigned)ploidy[s];
        std::vector<double> v;
        for (unsigned i = 1; i <= n; ++i) v.push_back(prob_m(i));
        return ln_one_minus_exp(ln_sum_exp(v));
    }
    static double binomial(unsigned n, unsigned k) {  // statrs factorial::binomial, exact for small n
        if (k > n) return 0.0;
        double r = 1.0;
        for (unsigned i = 1; i <= k; ++i) r = r * (double)(n - k + i) / (double)i;
        return std::floor(0.5 + r);
    }
    static double hypergeom_ln_pmf(unsigned N, unsigned K, unsigned n, unsigned x) {  // 584-598
        // statrs Hypergeometric::pmf: 0 outside [max(0, n+K-N), min(K, n)]
        unsigned lo = (n + K > N) ? (n + K - N) : 0, hi = std::min(K, n);
        if (x < lo || x > hi) return NEG_INF;
        double p = binomial(K, x) * binomial(N - K, n - x) / binomial(N, n);
        return std::log(p);
    }
    double prob_mendelian_alt_counts(unsigned sp0, unsigned sp1, unsigned tp, unsigned sa0, unsigned sa1, unsigned ta,
                                     double germline_rate_) const {  // 600-678
        auto cases = [](unsigned p) {
            std::vector<unsigned> v;
            if (p % 2 == 0) v.push_back(p / 2);
            else { v.push_back((unsigned)std::floor((double)p / 2.0)); v.push_back((unsigned)std::ceil((double)p / 2.0)); }
            return v;
        };
        bool valid = false;
        for (unsigned a : cases(sp0))
            for (unsigned b : cases(sp1))
                if (a + b == tp) valid = true;
        assert(valid && "ploidies of child and parents do not match");
        std::vector<double> probs;
        for (unsigned p1 : cases(sp0))
            for (unsigned p2 : cases(sp1)) {
                if (p1 + p2 != tp) continue;
                for (unsigned a1 = 0; a1 <= std::min(sa0, p1); ++a1)
                    for (unsigned a2 = 0; a2 <= std::min(sa1, p2); ++a2) {
                        if (a1 + a2 > ta) continue;
                        unsigned r1 = p1 - a1, r2 = p2 - a2;
                        double prob = hypergeom_ln_pmf(sp0, sa0, a1 + r1, a1) + hypergeom_ln_pmf(sp1, sa1, a2 + r2, a2);
                        int missing = (int)ta - (int)(a1 + a2);
                        probs.push_back(prob + std::log(germline_rate_) * (double)missing);
                    }
            }
        return ln_sum_exp(probs);
    }
    double prob_mendelian(int child, int p0, int p1, const std::vector<double>& ev, const std::vector<double>& g) const {  // 680-712
        auto n_alt = [&](int s) { return (unsigned)std::llround(g[s] * (double)ploidy[s]); };
        double gr = germline_rate[child] * variant_type_fraction();  // 259-261
        double prob = prob_mendelian_alt_counts((unsigned)ploidy[p0], (unsigned)ploidy[p1], (unsigned)ploidy[child], n_alt(p0),
                                                n_alt(p1), n_alt(child), gr);
        double r;
        if (somatic_rate_ln(child, &r)) prob += prob_somatic_mutation(r, eff_somatic(child, ev, g));
        return prob;
    }
    double calc_prob(const std::vector<double>& ev, std::vector<double> g) const {  // 298-438
        if ((int)g.size() == S) {
            double prob = 0.0, het;
            if (heterozygosity(&het)) {
                std::vector<int> pop;
                for (int s = 0; s < S; ++s)
                    if (inheritance[s].kind == VLR_INHERIT_NONE && ploidy[s] >= 0 && !uniform[s]) pop.push_back(s);
                prob = prob_population_germline(pop, g, het);
            }
            for (int s = 0; s < S; ++s) {
                if (uniform[s]) continue;
                const vlr_inheritance& inh = inheritance[s];
                if (inh.kind == VLR_INHERIT_MENDELIAN) prob += prob_mendelian(s, inh.from0, inh.from1, ev, g);
                else if (inh.kind == VLR_INHERIT_CLONAL) prob += prob_clonal(s, inh.from0, ev, g, inh.somatic != 0);
                else if (inh.kind == VLR_INHERIT_SUBCLONAL) prob += prob_subclonal(s, inh.from0, ev, g);
                else {
                    double r;
                    if (somatic_rate_ln(s, &r)) prob += prob_somatic_mutation(r, eff_somatic(s, ev, g));
                }
            }
            return prob;
        }
        int s = (int)g.size();
        auto push = [&](double germ) { std::vector<double> g2 = g; g2.push_back(germ); return g2; };
        if (ploidy[s] == 0 && ev[s] != 0.0) return NEG_INF;
        if (uniform[s]) {
            bool contained = false;
            for (auto& sp : universe[s])
                if (sp.contains(ev[s])) contained = true;
            if (contained) return calc_prob(ev, push(0.0));
            return NEG_INF;
        }
        if (!std::isnan(somatic_rate[s])) {
            assert(ploidy[s] >= 0);
            std::vector<double> probs;
            for (int n_alt = 0; n_alt <= ploidy[s]; ++n_alt) {
                double germ = ploidy[s] > 0 ? (double)n_alt / (double)ploidy[s] : 0.0;
                probs.push_back(calc_prob(ev, push(germ)));
            }
            return ln_sum_exp(probs);
        }
        if (ploidy[s] >= 0 && !std::isnan(heterozygosity_ln)) {
            if (is_valid_germline_vaf(s, ev[s])) return calc_prob(ev, push(ev[s]));
            return NEG_INF;
        }
        assert(false && "bug: not enough info for prior but no universe specified");
        return NEG_INF;
    }
    double compute(const Operands& ops) const {  // 715-762
        std::vector<double> ev(S);
        bool absent = true;
        for (int s = 0; s < S; ++s) {
            ev[s] = ops.events[s].af;
            if (ev[s] != 0.0) absent = false;
        }
        if (is_absent_only && !is_all_uniform()) {
            if (!absent) {
                double full = calc_prob(ev, {});
                if (full == NEG_INF) return full;
                std::vector<double> zero(S, 0.0);
                return ln_one_minus_exp(calc_prob(zero, {}));
            }
            return calc_prob(ev, {});
        }
        return calc_prob(ev, {});
    }
};

// ------------------------------------------------------------------ per-call context
struct Scenario {
    int S;
    std::vector<double> resolution;
    std::vector<int> contaminated_by;
    std::vector<double> purity_ln, impurity_ln;
    std::vector<Node> nodes;
    int n_named;
    std::vector<std::string> names;
    std::vector<std::vector<int>> named_roots;
    std::vector<int> absent_roots;  // VAFTree::absent (vaftree.rs:18-40)
    Prior prior;
};

struct JointEntry {
    Operands ops;
    double prob;
};

struct KeyHash {
    size_t operator()(const std::vector<uint64_t>& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint64_t v : k) { h ^= v; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

struct Ctx {
    const Scenario* sc;
    std::vector<Pileup> pileups;
    bool has_snv;
    uint8_t refbase, altbase;
    std::vector<Artifacts> hyps;  // learned; hyps[0] = none
    // likelihood caches (modes/generic.rs:38-53), fresh per Model::compute (bio Model)
    std::vector<std::unordered_map<std::vector<uint64_t>, double, KeyHash>> lik_cache;
    // joint_probs of bio's Model::compute
    std::unordered_map<std::vector<uint64_t>, size_t, KeyHash> joint_index;
    std::vector<JointEntry> joint;
    uint64_t n_lik_evals = 0, n_obs_terms = 0;
    bool nan_seen = false;
    // "tuned" CPU baseline (vlro_call_batch_tuned): the pileup likelihood in the affine form of SURVEY App. B — per observation
    // and hypothesis three linear-space coefficients, a pileup evaluation is a product of (c + q*alpha + e*beta) with the binary
    // exponent taken out every few terms and ONE logarithm at the end; no allocation in the term loop.  Tree walk, prior,
    // integrator and caches are the fidelity code above.  coef[s][h] is built on first use.
    bool tuned = false;
    struct Coef { bool built = false, slow = false, has_e = false; int group = 8; std::vector<double> c, q, e; };
    std::vector<std::vector<Coef>> coef;
};

inline uint64_t dbits(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }

std::vector<uint64_t> ops_key(const Operands& ops) {
    std::vector<uint64_t> k;
    for (auto& e : ops.events) {
        k.push_back(e.present ? 1 : 0);
        k.push_back(dbits(e.af));
        k.push_back((uint64_t)e.art * 2 + (e.discrete ? 1 : 0));
    }
    for (auto& l : ops.lfcs) {
        k.push_back(0xffff0000ull | ((uint64_t)l.sample_a << 8) | (uint64_t)l.sample_b);
        k.push_back((uint64_t)l.pred.cmp);
        k.push_back(dbits(l.pred.value));
    }
    return k;
}

// ------------------------------------------------------------------ likelihood (variants/model/likelihood.rs)
// likelihood.rs:43-53
inline double prob_sample_alt(const Obs& o, double ln_af) {
    if (ln_af != 0.0) {
        double p = ln_af + o.prob_sample_alt;
        // cap_numerical_overshoot(NUMERICAL_EPSILON = 1e-3): both addends <= 0, never triggers
        if (p > 0.0 && p <= 1e-3) p = 0.0;
        return p;
    }
    return ln_af;
}
// likelihood.rs:198-220
inline double likelihood_mapping(double ln_af, const Artifacts& a, const Obs& o) {
    double psa = prob_sample_alt(o, ln_af);
    double psr = ln_one_minus_exp(psa);
    double b_alt = art_prob_alt(a, o), b_ref = art_prob_ref(a, o);
    std::vector<double> v = {psa + b_alt + o.p_alt(), psr + o.p_ref() + b_ref};
    return ln_sum_exp(v);
}
// likelihood.rs:171-193
inline double lik_obs_single(double ln_af, const Artifacts& a, const Obs& o) {
    double prob = likelihood_mapping(ln_af, a, o);
    return ln_add_exp(o.prob_mapping + prob, o.prob_mismapping + o.prob_missed_allele + art_prob_any(a, o));
}
// likelihood.rs:86-115
inline double lik_obs_contaminated(double purity, double impurity, double ln_af_p, double ln_af_s, const Artifacts& ap,
                                   const Artifacts& as, const Obs& o) {
    double prob_primary = purity + likelihood_mapping(ln_af_p, ap, o);
    double prob_secondary = impurity + likelihood_mapping(ln_af_s, as, o);
    return ln_add_exp(o.prob_mapping + ln_add_exp(prob_secondary, prob_primary),
                      o.prob_mismapping + o.prob_missed_allele + art_prob_any(ap, o));
}

// ---- tuned baseline: affine coefficients (the same algebra as the engine's coefficient pass)
//   w = e^pm, u = e^(pmis + missed + b_any), A = e^(pa + b_alt), R = e^(pr + b_ref), s = e^prob_sample_alt
//   L_i(alpha, beta) = c + q*alpha + e*beta,  c = w R + u,  q = w s (A - R),  e = w (1 - s) (A - R)
//   single sample: alpha = a, beta = [a == 1];  contaminated: alpha = rho a + (1 - rho) a', beta = rho [a == 1] + (1 - rho) [a' == 1]
void build_coef(Ctx& c, int s, int h) {
    Ctx::Coef& k = c.coef[s][h];
    const Pileup& pile = c.pileups[s];
    const Artifacts& a = c.hyps[h];
    const size_t n = pile.obs.size();
    k.c.resize(n); k.q.resize(n); k.e.resize(n);
    double tmin = 1.0;
    bool bad = false;
    for (size_t i = 0; i < n; ++i) {
        const Obs& o = pile.obs[i];
        const double w = std::exp(o.prob_mapping);
        const double u = std::exp(o.prob_mismapping + o.prob_missed_allele + art_prob_any(a, o));
        const double A = std::exp(o.p_alt() + art_prob_alt(a, o)), R = std::exp(o.p_ref() + art_prob_ref(a, o));
        const double sa = std::exp(o.prob_sample_alt);
        const double d = w * (A - R);
        k.c[i] = w * R + u; k.q[i] = sa * d; k.e[i] = (1.0 - sa) * d;
        if (k.e[i] != 0.0) k.has_e = true;
        const double t0 = k.c[i], t1 = k.c[i] + k.q[i] + k.e[i], t2 = k.c[i] + k.q[i];
        const double m = std::min(t0, std::min(t1, t2));
        if (!(m > 0.0) || !std::isfinite(m) || !std::isfinite(t0 + t1 + t2)) bad = true;
        tmin = std::min(tmin, m);
    }
    // a term that is zero or denormal in linear space is still a finite logarithm in the reference: such pileups keep the
    // log-space code; terms above 2^-120 allow eight factors between two exponent extractions
    k.slow = bad || tmin < 1e-290;
    k.group = tmin >= 0x1p-120 ? 8 : 1;
    k.built = true;
}
inline double pileup_affine(const Ctx::Coef& k, double alpha, double beta) {
    const size_t n = k.c.size();
    const double* cc = k.c.data();
    const double* cq = k.q.data();
    const double* ce = k.e.data();
    const bool use_e = k.has_e && beta != 0.0;
    double P = 1.0;
    long E = 0;
    size_t i = 0;
    if (k.group == 8) {
        for (; i + 8 <= n; i += 8) {
            double p0 = 1.0, p1 = 1.0;
            if (use_e) {
                for (int j = 0; j < 8; j += 2) {
                    p0 *= std::fma(ce[i + j], beta, std::fma(cq[i + j], alpha, cc[i + j]));
                    p1 *= std::fma(ce[i + j + 1], beta, std::fma(cq[i + j + 1], alpha, cc[i + j + 1]));
                }
            } else {
                for (int j = 0; j < 8; j += 2) {
                    p0 *= std::fma(cq[i + j], alpha, cc[i + j]);
                    p1 *= std::fma(cq[i + j + 1], alpha, cc[i + j + 1]);
                }
            }
            int e0, e1;
            const double m = std::frexp(P, &e0) * std::frexp(p0 * p1, &e1);
            P = m;
            E += e0 + e1;
        }
    }
    for (; i < n; ++i) {
        const double t = use_e ? std::fma(ce[i], beta, std::fma(cq[i], alpha, cc[i])) : std::fma(cq[i], alpha, cc[i]);
        int e0, e1;
        P = std::frexp(P, &e0) * std::frexp(t, &e1);
        E += e0 + e1;
    }
    return std::log(P) + (double)E * 0.6931471805599453;
}

// modes/generic.rs:496-555 GenericLikelihood::compute
double likelihood_compute(Ctx& c, const Operands& ops) {
    const Scenario& sc = *c.sc;
    for (auto& l : ops.lfcs) {  // 503-509
        if (!l.pred.is_true(ops.events[l.sample_a].af, ops.events[l.sample_b].af)) return NEG_INF;
    }
    double p = 0.0;
    for (int s = 0; s < sc.S; ++s) {  // 514-551
        const SampleEvent& e = ops.events[s];
        const Pileup& pile = c.pileups[s];
        int by = sc.contaminated_by[s];
        std::vector<uint64_t> key = {dbits(e.af), (uint64_t)e.art * 2 + e.discrete};
        if (by >= 0) {
            const SampleEvent& e2 = ops.events[by];
            key.push_back(dbits(e2.af));
            key.push_back((uint64_t)e2.art * 2 + e2.discrete);
        }
        auto it = c.lik_cache[s].find(key);
        double lh;
        if (it != c.lik_cache[s].end()) {
            lh = it->second;
        } else {
            lh = 0.0;
            c.n_lik_evals++;
            c.n_obs_terms += pile.obs.size();
            bool done = false;
            if (c.tuned && (by < 0 || ops.events[by].art == e.art)) {
                if (!c.coef[s][e.art].built) build_coef(c, s, e.art);
                const Ctx::Coef& k = c.coef[s][e.art];
                if (!k.slow) {
                    double alpha = e.af, beta = e.af == 1.0 ? 1.0 : 0.0;
                    if (by >= 0) {
                        const double rho = std::exp(sc.purity_ln[s]), af2 = ops.events[by].af;
                        alpha = rho * e.af + (1.0 - rho) * af2;
                        beta = rho * beta + (1.0 - rho) * (af2 == 1.0 ? 1.0 : 0.0);
                    }
                    lh = pileup_affine(k, alpha, beta);
                    done = true;
                }
            }
            if (done) {
            } else if (by >= 0) {  // likelihood.rs:122-157
                const SampleEvent& e2 = ops.events[by];
                double la = std::log(e.af), lb = std::log(e2.af);
                for (auto& o : pile.obs)
                    lh += lik_obs_contaminated(sc.purity_ln[s], sc.impurity_ln[s], la, lb, c.hyps[e.art], c.hyps[e2.art], o);
            } else {  // likelihood.rs:227-249
                double la = std::log(e.af);
                for (auto& o : pile.obs) lh += lik_obs_single(la, c.hyps[e.art], o);
            }
            if (std::isnan(lh)) c.nan_seen = true;
            c.lik_cache[s][key] = lh;
        }
        p += lh;
    }
    return p;
}

// joint_prob closure of bio's Model::compute: prior + likelihood, recorded in joint_probs
double joint_prob(Ctx& c, const Operands& ops) {
    double p = c.sc->prior.compute(ops) + likelihood_compute(c, ops);
    auto key = ops_key(ops);
    auto it = c.joint_index.find(key);
    if (it == c.joint_index.end()) {
        c.joint_index[key] = c.joint.size();
        c.joint.push_back({ops, p});
    } else {
        c.joint[it->second].prob = p;
    }
    return p;
}

// ------------------------------------------------------------------ adaptive integration (utils/adaptive_integration.rs:25-141)
template <typename F>
double ln_integrate_exp(F&& density, double min_point, double max_point, double max_resolution) {
    std::map<double, double> probs;
    auto grid_point = [&](double x) { probs[x] = density(x); return x; };
    auto mid_of = [](double l, double r) { return (r + l) / 2.0; };
    double left = grid_point(min_point);
    double right = grid_point(max_point);
    bool have_middle = false, have_first = false;
    double middle = 0.0, first_middle = 0.0;
    while ((((right - left) >= max_resolution) && left < right) || !have_middle) {
        middle = grid_point(mid_of(left, right));
        have_middle = true;
        double middle1 = grid_point(mid_of(left, middle));
        double middle2 = grid_point(mid_of(middle, right));
        if (!have_first) { first_middle = middle; have_first = true; }
        double xs[4] = {left, middle1, middle2, right};
        // argmax over {left, middle1, middle2, right}; reference iterates a HashMap (ties unspecified):
        // deterministic rule = strictly-greater update in index order, i.e. lowest index wins ties.
        int k = 0;
        for (int i = 1; i < 4; ++i)
            if (probs[xs[i]] > probs[xs[k]]) k = i;
        left = (k > 0) ? xs[k - 1] : xs[k];
        right = (k < 3) ? xs[k + 1] : xs[k];
    }
    // abandoned arm (95-106)
    if (middle < first_middle) grid_point(mid_of(first_middle, max_point));
    else grid_point(mid_of(min_point, first_middle));
    // small interval around the optimum (107-131)
    {
        double lo = std::max(middle - max_resolution * 3.0, min_point);
        double hi = std::min(middle + max_resolution * 3.0, max_point);
        std::vector<double> a = linspace(lo, middle, 4), b = linspace(middle, hi, 4);
        for (int i = 0; i < 3; ++i) grid_point(a[i]);
        for (int i = 1; i < 4; ++i) grid_point(b[i]);
    }
    // ln_trapezoidal_integrate_grid_exp over the sorted visited points (133-140; bio)
    std::vector<double> g, f;
    for (auto& kv : probs) { g.push_back(kv.first); f.push_back(kv.second); }
    std::vector<double> parts;
    for (size_t i = 0; i + 1 < g.size(); ++i) parts.push_back(ln_add_exp(f[i], f[i + 1]) + std::log((g[i + 1] - g[i]) / 2.0));
    return ln_sum_exp(parts);
}
// bio LogProb::ln_simpsons_integrate_exp
template <typename F>
double ln_simpsons_integrate_exp(F&& density, double a, double b, int n) {
    assert(n % 2 == 1);
    std::vector<double> grid = linspace(a, b, n);
    std::vector<double> probs;
    for (int i = 1; i < n - 1; ++i) {
        double weight = (double)(2 + (i % 2) * 2);
        probs.push_back(density(grid[i]) + std::log(weight));
    }
    probs.push_back(density(a));
    probs.push_back(density(b));
    double width = b - a;
    return ln_sum_exp(probs) + std::log(width) - std::log((double)(n - 1)) - std::log(3.0);
}

// ------------------------------------------------------------------ posterior (modes/generic.rs:190-461)
// LikelihoodOperands::lfc_bounds (modes/generic.rs:148-174)
bool lfc_bounds(const Operands& ops, int sample, Range* out) {
    bool have = false;
    Range acc = Range::empty();
    for (auto& l : ops.lfcs) {
        bool got = false;
        Range b = Range::empty();
        if (l.sample_a == sample) {
            if (ops.events[l.sample_b].present) { b = l.pred.invert().infer_vaf_bounds(ops.events[l.sample_b].af); got = true; }
        } else if (l.sample_b == sample) {
            if (ops.events[l.sample_a].present) { b = l.pred.infer_vaf_bounds(ops.events[l.sample_a].af); got = true; }
        }
        if (got) {
            acc = have ? acc.intersect(b) : b;
            have = true;
        }
    }
    *out = acc;
    return have;
}

double density(Ctx& c, int node_id, Operands& ops, int bias);

double subdensity(Ctx& c, const Node& node, Operands& ops, int bias) {  // 199-230
    double p;
    if (node.children.empty()) {
        p = joint_prob(c, ops);
    } else if (node.children.size() > 1) {
        std::vector<double> v;
        for (int ch : node.children) {
            Operands cp = ops;
            v.push_back(density(c, ch, cp, bias));
        }
        p = ln_sum_exp(v);
    } else {
        p = density(c, node.children[0], ops, bias);
    }
    if (std::isnan(p)) c.nan_seen = true;
    return p;
}

double density(Ctx& c, int node_id, Operands& ops, int bias) {
    const Scenario& sc = *c.sc;
    const Node& node = sc.nodes[node_id];
    switch (node.kind) {
        case VLR_NODE_LFC:  // 233-244
            ops.lfcs.push_back({node.sample, node.sample_b, node.pred});
            return subdensity(c, node, ops, bias);
        case VLR_NODE_FALSE: return NEG_INF;
        case VLR_NODE_TRUE: return 0.0;
        case VLR_NODE_VARIANT: {  // 398-420
            if (c.has_snv) {
                bool contains = iupac_contains(node.refbase, c.refbase) && iupac_contains(node.altbase, c.altbase);
                if ((node.positive && !contains) || (!node.positive && contains)) return NEG_INF;
                return subdensity(c, node, ops, bias);
            } else if (node.positive) {
                return NEG_INF;
            }
            return subdensity(c, node, ops, bias);
        }
        default: break;
    }
    // Sample node (247-397)
    int sample = node.sample;
    auto push_base_event = [&](double af, Operands& o, bool discrete) {
        SampleEvent e;
        e.present = true; e.af = af; e.art = bias; e.discrete = discrete;
        o.events[sample] = e;
    };
    Range bounds;
    bool have_bounds = lfc_bounds(ops, sample, &bounds);
    if (have_bounds && bounds.is_empty()) return NEG_INF;  // 262-268
    const Pileup& pile = c.pileups[sample];
    size_t n_obs = pile.obs.size();  // 270-291 (no depth observations in format v15: preprocessing/mod.rs:913)
    bool is_clear_ref = n_obs > 10;
    if (is_clear_ref)
        for (auto& o : pile.obs)
            if (!o.is_positive_ref_support()) { is_clear_ref = false; break; }

    if (node.vafs.is_set) {  // 294-330
        bool all_pos = true;
        for (double v : node.vafs.set)
            if (!(v > 0.0)) all_pos = false;
        if (is_clear_ref && all_pos) return NEG_INF;
        std::vector<double> vafs;
        for (double v : node.vafs.set)
            if (!have_bounds || bounds.contains(v)) vafs.push_back(v);
        if (vafs.size() == 1) {
            push_base_event(vafs[0], ops, true);
            return subdensity(c, node, ops, bias);
        }
        std::vector<double> vals;
        for (double v : vafs) {
            Operands cp = ops;
            push_base_event(v, cp, true);
            vals.push_back(subdensity(c, node, cp, bias));
        }
        return ln_sum_exp(vals);
    }
    // Range (331-395)
    Range vafs = have_bounds ? node.vafs.range.intersect(bounds) : node.vafs.range;
    if (vafs.is_empty()) return NEG_INF;
    if (is_clear_ref && vafs.start > 0.0) return NEG_INF;
    if (vafs.is_singleton()) {
        push_base_event(vafs.start, ops, true);
        return subdensity(c, node, ops, bias);
    }
    double resolution = sc.resolution[sample];
    double min_vaf = vafs.observable_min(n_obs);
    double max_vaf = vafs.observable_max(n_obs);
    assert(min_vaf <= max_vaf);
    auto dens = [&](double vaf) {
        Operands cp = ops;
        push_base_event(vaf, cp, false);
        return subdensity(c, node, cp, bias);
    };
    if ((max_vaf - min_vaf) < resolution) return ln_simpsons_integrate_exp(dens, min_vaf, max_vaf, 3);
    if (n_obs < 5) return ln_simpsons_integrate_exp(dens, min_vaf, max_vaf, 11);
    return ln_integrate_exp(dens, min_vaf, max_vaf, resolution);
}

// vaftree.rs:42-51, 116-164
bool node_contains(const Scenario& sc, int node_id, const Operands& ops, std::vector<const VafLfc*>& lfcs, int exclude) {
    const Node& node = sc.nodes[node_id];
    bool contained;
    switch (node.kind) {
        case VLR_NODE_SAMPLE:
            if (exclude == node.sample) return true;
            contained = node.vafs.contains(ops.events[node.sample].af);
            break;
        case VLR_NODE_LFC: {
            bool found_any = false;
            std::vector<const VafLfc*> keep;
            for (auto* l : lfcs) {
                bool found = l->sample_a == node.sample && l->sample_b == node.sample_b && l->pred == node.pred;
                found_any |= found;
                if (!found) keep.push_back(l);
            }
            lfcs = keep;
            contained = found_any;
            break;
        }
        case VLR_NODE_FALSE: contained = false; break;
        default: contained = true; break;  // True, Variant
    }
    if (node.children.empty()) return contained && lfcs.empty();
    if (!contained) return false;
    for (int ch : node.children) {
        if (node.children.size() == 1) {
            if (node_contains(sc, ch, ops, lfcs, exclude)) return true;
        } else {
            std::vector<const VafLfc*> cp = lfcs;
            if (node_contains(sc, ch, ops, cp, exclude)) return true;
        }
    }
    return false;
}
bool tree_contains(const Scenario& sc, const std::vector<int>& roots, const Operands& ops, int exclude) {
    for (int r : roots) {
        std::vector<const VafLfc*> lfcs;
        for (auto& l : ops.lfcs) lfcs.push_back(&l);
        if (node_contains(sc, r, ops, lfcs, exclude)) return true;
    }
    return false;
}

// ------------------------------------------------------------------ scenario construction
void build_scenario(const vlr_scenario_desc* d, Scenario& sc) {
    sc.S = d->n_samples;
    int S = sc.S;
    sc.resolution.assign(d->resolution, d->resolution + S);
    sc.contaminated_by.assign(d->contaminated_by, d->contaminated_by + S);
    sc.purity_ln.assign(S, 0.0);
    sc.impurity_ln.assign(S, NEG_INF);
    for (int s = 0; s < S; ++s)
        if (sc.contaminated_by[s] >= 0) {
            double purity = 1.0 - d->contamination_fraction[s];  // modes/generic.rs:482-484
            assert(purity > 0.0 && purity <= 1.0);              // likelihood.rs:78
            sc.purity_ln[s] = std::log(purity);
            sc.impurity_ln[s] = ln_one_minus_exp(sc.purity_ln[s]);
        }
    sc.nodes.resize(d->n_nodes);
    for (int i = 0; i < d->n_nodes; ++i) {
        const vlr_node& n = d->nodes[i];
        Node& o = sc.nodes[i];
        o.kind = n.kind;
        o.sample = n.sample;
        o.sample_b = n.sample_b;
        o.pred = {n.cmp, n.lfc_value};
        if (n.kind == VLR_NODE_SAMPLE) o.vafs = spectrum_from(n.vafs, d->vafs);
        o.positive = n.positive != 0;
        o.refbase = n.refbase;
        o.altbase = n.altbase;
        o.children.assign(d->child_index + n.child_offset, d->child_index + n.child_offset + n.n_children);
    }
    sc.n_named = d->n_events;
    for (int e = 0; e < d->n_events; ++e) {
        sc.names.push_back(d->event_names ? d->event_names[e] : ("event" + std::to_string(e)));
        sc.named_roots.push_back(std::vector<int>(d->root_index + d->event_root_offset[e], d->root_index + d->event_root_offset[e + 1]));
    }
    // VAFTree::absent(n_samples): chain sample 0 -> 1 -> ... each {0.0} (vaftree.rs:18-40)
    int base = (int)sc.nodes.size();
    for (int s = 0; s < S; ++s) {
        Node n;
        n.kind = VLR_NODE_SAMPLE;
        n.sample = s;
        n.vafs.is_set = true;
        n.vafs.set = {0.0};
        n.vafs.range = Range::empty();
        if (s + 1 < S) n.children.push_back(base + s + 1);
        sc.nodes.push_back(n);
    }
    sc.absent_roots = {base};
    Prior& p = sc.prior;
    p.S = S;
    p.uniform.assign(d->uniform_prior, d->uniform_prior + S);
    p.ploidy.assign(d->ploidy, d->ploidy + S);
    p.germline_rate.assign(d->germline_mutation_rate, d->germline_mutation_rate + S);
    p.somatic_rate.assign(d->somatic_effective_mutation_rate, d->somatic_effective_mutation_rate + S);
    p.inheritance.assign(d->inheritance, d->inheritance + S);
    p.heterozygosity_ln = std::isnan(d->heterozygosity) ? NAN : std::log(d->heterozygosity);  // calling.rs:1079-1084
    p.variant_heterozygosity_ln = d->variant_heterozygosity_ln;
    p.variant_somatic_rate_ln = d->variant_somatic_effective_mutation_rate_ln;
    p.frac_indel = d->fraction_indel;
    p.frac_mnv = d->fraction_mnv;
    p.frac_sv = d->fraction_sv;
    p.is_absent_only = d->is_absent_only != 0;
    p.universe.resize(S);
    for (int s = 0; s < S; ++s)
        for (int i = d->universe_offset[s]; i < d->universe_offset[s + 1]; ++i) p.universe[s].push_back(spectrum_from(d->universe[i], d->vafs));
}

Obs make_obs(const vlr_batch* b, int64_t i) {
    Obs o;
    o.prob_mapping = b->prob_mapping[i];
    o.prob_mismapping = ln_one_minus_exp(o.prob_mapping);  // read_observation.rs:283-286
    o.prob_alt = b->prob_alt[i];
    o.prob_ref = b->prob_ref[i];
    o.prob_missed_allele = b->prob_missed_allele[i];
    o.prob_sample_alt = b->prob_sample_alt[i];
    o.prob_double_overlap = b->prob_double_overlap[i];
    o.prob_single_overlap = ln_one_minus_exp(o.prob_double_overlap);  // 288-291
    o.prob_hit_base = b->prob_hit_base[i];
    uint32_t f = b->flags[i];
    o.strand = (f >> VLR_F_STRAND_SHIFT) & 3;
    o.orientation = (f >> VLR_F_ORIENT_SHIFT) & 3;
    o.readpos_major = (f & VLR_F_READPOS_MAJOR) != 0;
    o.softclipped = (f & VLR_F_SOFTCLIPPED) != 0;
    o.paired = (f & VLR_F_PAIRED) != 0;
    o.is_max_mapq = (f & VLR_F_MAX_MAPQ) != 0;
    o.alt_locus = (f >> VLR_F_ALTLOCUS_SHIFT) & 3;
    o.has_hp_len = (f & VLR_F_HP_LEN_VALID) != 0;
    o.hp_len = (int)(int8_t)((f >> VLR_F_HP_LEN_SHIFT) & 0xff);
    o.has_hp_art = b->prob_hp_artifact && !std::isnan(b->prob_hp_artifact[i]);
    o.has_hp_var = b->prob_hp_variant && !std::isnan(b->prob_hp_variant[i]);
    o.hp_art = o.has_hp_art ? b->prob_hp_artifact[i] : 0.0;
    o.hp_var = o.has_hp_var ? b->prob_hp_variant[i] : 0.0;
    return o;
}

// MAP ordering: reference sorts joint_probs by posterior descending with unspecified tie order
// (bio ModelInstance::event_posteriors over a HashMap).  Deterministic rule: higher prob first;
// ties: non-artifact first, then lower hypothesis id, then lexicographically smaller VAF tuple, then the operand
// set whose first differing is_discrete flag is set.
bool map_before(const JointEntry& a, const JointEntry& b) {
    if (a.prob != b.prob) return a.prob > b.prob;
    int aa = 0, ab = 0;
    for (auto& e : a.ops.events) aa = std::max(aa, e.art);
    for (auto& e : b.ops.events) ab = std::max(ab, e.art);
    if (aa != ab) return aa < ab;
    for (size_t s = 0; s < a.ops.events.size(); ++s)
        if (a.ops.events[s].af != b.ops.events[s].af) return a.ops.events[s].af < b.ops.events[s].af;
    for (size_t s = 0; s < a.ops.events.size(); ++s)
        if (a.ops.events[s].discrete != b.ops.events[s].discrete) return a.ops.events[s].discrete;
    return false;
}

}  // namespace

// ====================================================================== exported C interface
extern "C" {

typedef struct {
    uint64_t n_lik_evals;  // pileup-likelihood evaluations (cache misses), summed over loci
    uint64_t n_obs_terms;  // observation terms evaluated
    uint64_t n_joint;      // distinct visited operands
} vlro_stats;

// Evaluate loci [locus_begin, locus_end) of a HOST batch with the restated reference algorithm
// (Caller::call_record, calling.rs:720-842 incl. preprocess_record's pileup edits 590-625).
// `event_ln_posterior` (optional) receives [n * (1 + 2*n_events)] posteriors of the full event universe
// (absent, then clean/artifact twin per scenario event; -inf-filled twin columns when no bias is enabled).
static int call_batch_impl(const vlr_scenario_desc* desc, const vlr_batch* in, vlr_results* out, int64_t locus_begin,
                           int64_t locus_end, double* event_ln_posterior, vlro_stats* stats, bool tuned);
int vlro_call_batch(const vlr_scenario_desc* desc, const vlr_batch* in, vlr_results* out, int64_t locus_begin,
                    int64_t locus_end, double* event_ln_posterior, vlro_stats* stats) {
    return call_batch_impl(desc, in, out, locus_begin, locus_end, event_ln_posterior, stats, false);
}
// The tuned CPU baseline of bench.py (cpu_baseline.kind = "tuned"): same tree walk, prior, integrator and result extraction, the
// pileup likelihood in affine product form (Ctx::tuned).  Not bit-identical to the fidelity path (products instead of sums of
// logarithms); bench.py reports its deviation.
int vlro_call_batch_tuned(const vlr_scenario_desc* desc, const vlr_batch* in, vlr_results* out, int64_t locus_begin,
                          int64_t locus_end, double* event_ln_posterior, vlro_stats* stats) {
    return call_batch_impl(desc, in, out, locus_begin, locus_end, event_ln_posterior, stats, true);
}
static int call_batch_impl(const vlr_scenario_desc* desc, const vlr_batch* in, vlr_results* out, int64_t locus_begin,
                           int64_t locus_end, double* event_ln_posterior, vlro_stats* stats, bool tuned) {
    Scenario sc;
    build_scenario(desc, sc);
    const int S = sc.S;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const int n_out = sc.n_named + 2;
    const int n_univ = 1 + 2 * sc.n_named;

    for (int64_t l = locus_begin; l < locus_end; ++l) {
        Ctx c;
        c.sc = &sc;
        uint8_t lf = in->locus_flags[l];
        c.has_snv = (lf & VLR_LOCUS_HAS_SNV) != 0;
        c.refbase = in->ref_base ? in->ref_base[l] : 0;
        c.altbase = in->alt_base ? in->alt_base[l] : 0;
        sc.prior.variant_type = in->variant_type ? (int)in->variant_type[l] : (int)VLR_VT_SNV;
        uint32_t status = 0;

        // ---- preprocess_record (calling.rs:581-625)
        c.pileups.resize(S);
        bool filtered = false;
        for (int s = 0; s < S; ++s) {
            int64_t p = l * S + s;
            for (int64_t i = in->obs_offset[p]; i < in->obs_offset[p + 1]; ++i) {
                Obs o = make_obs(in, i);
                if (lf & VLR_LOCUS_REMOVE_NONSTANDARD) {  // pileup.rs:26-43
                    if (o.orientation == VLR_ORIENT_OTHER) { c.pileups[s].n_filtered++; continue; }
                }
                c.pileups[s].obs.push_back(o);
            }
            if (c.pileups[s].n_filtered > 0) filtered = true;
        }
        {  // adjust_singleton_evidence (read_observation.rs:548-562)
            Obs* single = nullptr;
            int n_alt = 0;
            for (auto& p : c.pileups)
                for (auto& o : p.obs)
                    if (o.prob_alt > o.prob_ref) { n_alt++; single = &o; }
            if (n_alt == 1) {
                single->has_adj = true;
                single->prob_alt_adj = LN_05;
                single->prob_ref_adj = LN_05;
                status |= VLR_LOCUS_SINGLETON_ADJ;
            }
        }
        if (filtered) status |= VLR_LOCUS_FILTERED_ALN;
        bool missing = true;
        for (auto& p : c.pileups)
            if (!p.obs.empty()) missing = false;
        if (missing) status |= VLR_LOCUS_MISSING_DATA;

        // ---- event universe (calling.rs:654-687) + learn_parameters (749-757)
        c.hyps.clear();
        c.hyps.push_back(Artifacts());
        for (auto& a : all_artifact_combinations(lf & 0x3f)) c.hyps.push_back(a);
        for (size_t h = 0; h < c.hyps.size(); ++h) {
            art_learn(c.hyps[h], c.pileups);
            c.hyps[h].id = (int)h;
        }
        std::vector<Event> universe;
        {
            Event e;
            e.name = "absent"; e.roots = sc.absent_roots; e.biases = {0}; e.artifact = false; e.named_index = -1;
            universe.push_back(e);
        }
        for (int n = 0; n < sc.n_named; ++n) {
            Event e;
            e.name = sc.names[n]; e.roots = sc.named_roots[n]; e.biases = {0}; e.artifact = false; e.named_index = n;
            universe.push_back(e);
            if (c.hyps.size() > 1) {
                Event t = e;
                t.biases.clear();
                for (size_t h = 1; h < c.hyps.size(); ++h) t.biases.push_back((int)h);
                t.artifact = true;
                universe.push_back(t);
            }
        }
        c.lik_cache.assign(S, {});
        c.tuned = tuned;
        if (tuned) c.coef.assign(S, std::vector<Ctx::Coef>(c.hyps.size()));

        // ---- Model::compute (bio) with GenericPosterior::compute (modes/generic.rs:425-461)
        std::vector<double> value(universe.size());
        for (size_t ei = 0; ei < universe.size(); ++ei) {
            const Event& ev = universe[ei];
            double bias_prior = ev.artifact ? LN_05 + std::log(1.0 / (double)ev.biases.size()) : LN_05;
            std::vector<double> terms;
            for (int h : ev.biases) {
                if (!art_gate(c.hyps[h], c.pileups)) continue;
                for (int root : ev.roots) {
                    Operands ops;
                    ops.events.assign(S, SampleEvent());
                    terms.push_back(bias_prior + density(c, root, ops, h));
                }
            }
            value[ei] = ln_sum_exp(terms);
        }
        double marginal = ln_sum_exp(value);
        std::vector<double> post(universe.size());
        for (size_t ei = 0; ei < universe.size(); ++ei) post[ei] = value[ei] - marginal;
        if (c.nan_seen || std::isnan(marginal)) status |= VLR_LOCUS_NAN;

        // ---- call_record post-processing (calling.rs:762-803)
        // best event = minmax_by_key max: last maximal element in universe order
        size_t best = 0;
        for (size_t ei = 1; ei < universe.size(); ++ei)
            if (!(post[ei] < post[best])) best = ei;  // >= : later index wins ties (itertools minmax)
        std::vector<double> art_terms;
        for (size_t ei = 0; ei < universe.size(); ++ei)
            if (universe[ei].artifact) art_terms.push_back(post[ei]);
        double prob_artifact = ln_sum_exp(art_terms);
        bool is_artifact = true;
        for (size_t ei = 0; ei < universe.size(); ++ei)
            if (!universe[ei].artifact && !(post[ei] < prob_artifact)) is_artifact = false;

        double* lp = out->ln_posterior + l * n_out;
        for (int k = 0; k < n_out; ++k) lp[k] = NEG_INF;
        for (size_t ei = 0; ei < universe.size(); ++ei)
            if (!universe[ei].artifact) lp[universe[ei].named_index + 1] = post[ei];
        lp[n_out - 1] = prob_artifact;
        if (out->ln_marginal) out->ln_marginal[l] = marginal;
        if (event_ln_posterior) {
            double* ep = event_ln_posterior + (l - locus_begin) * n_univ;
            for (int k = 0; k < n_univ; ++k) ep[k] = NEG_INF;
            for (size_t ei = 0; ei < universe.size(); ++ei) {
                int col = universe[ei].named_index < 0 ? 0 : 1 + 2 * universe[ei].named_index + (universe[ei].artifact ? 1 : 0);
                ep[col] = post[ei];
            }
        }
        if (out->best_event) {
            const Event& be = universe[best];
            out->best_event[l] = be.named_index < 0 ? 0 : 1 + 2 * be.named_index + (be.artifact ? 1 : 0);
        }

        // ---- sample_infos (calling.rs:844-937)
        std::vector<size_t> order(c.joint.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return map_before(c.joint[a], c.joint[b]); });
        const JointEntry* map = nullptr;
        for (size_t idx : order) {
            const JointEntry& je = c.joint[idx];
            bool any_art = false;
            for (auto& e : je.ops.events)
                if (c.hyps[e.art].is_artifact()) any_art = true;
            if (any_art && !is_artifact) continue;
            if (!tree_contains(sc, universe[best].roots, je.ops, -1)) continue;
            map = &je;
            break;
        }
        for (int s = 0; s < S; ++s) out->map_vaf[l * S + s] = NAN;
        if (out->map_bias) std::memset(out->map_bias + l * VLR_N_BIAS, 0, VLR_N_BIAS);
        if (out->afd_count)
            for (int s = 0; s < S; ++s) out->afd_count[l * S + s] = 0;
        if (map) {
            for (int s = 0; s < S; ++s) {
                const SampleEvent& est = map->ops.events[s];
                const Artifacts& a = c.hyps[est.art];
                if (a.is_artifact()) {
                    out->map_vaf[l * S + s] = 0.0;
                    if (out->map_bias) {
                        uint8_t* mb = out->map_bias + l * VLR_N_BIAS;
                        mb[0] = (uint8_t)a.sb; mb[1] = (uint8_t)a.rob; mb[2] = (uint8_t)a.rpb; mb[3] = (uint8_t)a.scb;
                        mb[4] = (uint8_t)a.he; mb[5] = (uint8_t)a.alb;
                    }
                } else {
                    out->map_vaf[l * S + s] = est.af;
                }
                // AFD (calling.rs:889-928)
                if (out->afd_count && !a.is_artifact()) {
                    std::vector<std::pair<double, double>> dist;
                    for (size_t idx : order) {
                        const JointEntry& je = c.joint[idx];
                        if (!tree_contains(sc, universe[best].roots, je.ops, s)) continue;
                        const SampleEvent& e = je.ops.events[s];
                        if (c.hyps[e.art].is_artifact()) continue;
                        bool others = true;
                        for (int o2 = 0; o2 < S; ++o2) {
                            if (o2 == s) continue;
                            const SampleEvent& x = je.ops.events[o2];
                            const SampleEvent& m = map->ops.events[o2];
                            if (!(x.af == m.af && x.art == m.art && x.discrete == m.discrete)) others = false;
                        }
                        if (!others) continue;
                        dist.push_back({e.af, je.prob - marginal});
                    }
                    std::stable_sort(dist.begin(), dist.end(), [](const std::pair<double, double>& x, const std::pair<double, double>& y) { return x.first < y.first; });
                    int n = (int)std::min<size_t>(dist.size(), (size_t)out->afd_capacity);
                    out->afd_count[l * S + s] = (int)dist.size();
                    for (int i = 0; i < n; ++i) {
                        out->afd_vaf[(l * S + s) * out->afd_capacity + i] = dist[i].first;
                        out->afd_lnprob[(l * S + s) * out->afd_capacity + i] = dist[i].second;
                    }
                }
            }
        }
        out->status[l] = status;
        if (stats) {
            stats->n_lik_evals += c.n_lik_evals;
            stats->n_obs_terms += c.n_obs_terms;
            stats->n_joint += c.joint.size();
        }
    }
    return 0;
}

// ---- small probes for unit tests of the restated primitives (tests/test_oracle_units.py)
double vlro_ln_one_minus_exp(double p) { return ln_one_minus_exp(p); }
double vlro_ln_add_exp(double a, double b) { return ln_add_exp(a, b); }
double vlro_ln_sum_exp(const double* v, int n) { return ln_sum_exp(std::vector<double>(v, v + n)); }
double vlro_observable_min(double start, double end, int lex, int rex, int n) { return Range{start, end, lex != 0, rex != 0}.observable_min((size_t)n); }
double vlro_observable_max(double start, double end, int lex, int rex, int n) { return Range{start, end, lex != 0, rex != 0}.observable_max((size_t)n); }
// adaptive integration of exp(-(x-mu)^2 / (2 s^2)) for a self-check against quadrature
double vlro_adaptive_gauss(double lo, double hi, double res, double mu, double sigma, int* n_points) {
    int n = 0;
    double r = ln_integrate_exp([&](double x) { n++; return -(x - mu) * (x - mu) / (2 * sigma * sigma); }, lo, hi, res);
    if (n_points) *n_points = n;
    return r;
}
// prior probe: ln prior of a VAF tuple under a scenario (variants/model/prior.rs:715-762)
double vlro_prior(const vlr_scenario_desc* desc, const double* vafs, int variant_type) {
    Scenario sc;
    build_scenario(desc, sc);
    sc.prior.variant_type = variant_type;
    Operands ops;
    ops.events.assign(sc.S, SampleEvent());
    for (int s = 0; s < sc.S; ++s) { ops.events[s].present = true; ops.events[s].af = vafs[s]; ops.events[s].discrete = true; }
    return sc.prior.compute(ops);
}
// single-observation likelihood probes (likelihood.rs:171-193, 86-115) with Artifacts::none() learned at rate .5
double vlro_lik_obs_single(const vlr_batch* b, int64_t i, double af) {
    Obs o = make_obs(b, i);
    Artifacts a;
    return lik_obs_single(std::log(af), a, o);
}
double vlro_lik_obs_contaminated(const vlr_batch* b, int64_t i, double purity, double af_p, double af_s) {
    Obs o = make_obs(b, i);
    Artifacts a;
    double lp = std::log(purity);
    return lik_obs_contaminated(lp, ln_one_minus_exp(lp), std::log(af_p), std::log(af_s), a, a, o);
}
// pileup likelihood probes (likelihood.rs:122-157, 227-249) over observations [i0, i1) with Artifacts::none()
double vlro_pileup_lik_single(const vlr_batch* b, int64_t i0, int64_t i1, double af) {
    Artifacts a;
    double lh = 0.0, la = std::log(af);
    for (int64_t i = i0; i < i1; ++i) lh += lik_obs_single(la, a, make_obs(b, i));
    return lh;
}
double vlro_pileup_lik_contaminated(const vlr_batch* b, int64_t i0, int64_t i1, double purity, double af_p, double af_s) {
    Artifacts a;
    double lp = std::log(purity), li = ln_one_minus_exp(lp), la = std::log(af_p), lb = std::log(af_s), lh = 0.0;
    for (int64_t i = i0; i < i1; ++i) lh += lik_obs_contaminated(lp, li, la, lb, a, a, make_obs(b, i));
    return lh;
}
double vlro_bias_prob_ref_none(const vlr_batch* b, int64_t i) {
    Obs o = make_obs(b, i);
    Artifacts a;
    return art_prob_ref(a, o);
}

}  // extern "C"
