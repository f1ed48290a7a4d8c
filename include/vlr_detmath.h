/*
 * vlr_detmath.h — platform-independent exp / log1p / log2-ratio for the *decision* arithmetic (bias gating, l2fc predicates).
 *
 * Why: StrandBias::estimate_forward_rate (reference src/variants/model/bias/strand_bias.rs:79-123) and
 * ReadPositionBias::has_valid_major_rate (read_position_bias.rs:63-122) compare ratios of
 * exp(ln_sum_exp(prob_mapping...)) against literal thresholds (0.4, 0.6, 2, 10, 100, 0.05).  prob_mapping
 * is constant within a pileup (MAPQ adjustment, read_observation.rs:456-502), so the ratio is
 * mathematically k/n and e.g. 3 of 5 forward reads sits EXACTLY on the 0.6 boundary: the outcome then
 * depends on the last-bit rounding of libm's exp/log1p, i.e. on the platform the reference binary runs
 * on (glibc selects FMA/non-FMA variants at run time).  To make the CPU oracle and the GPU kernel take
 * identical decisions there, both evaluate these few expressions with the functions below, which use
 * only IEEE-754 +,-,*,/ and fma in a fixed order (bit-identical on x86-64 and gfx950; ~1-2 ulp).
 * Everything else (likelihoods, integrals) uses the platform libm: there 1e-16 differences are harmless.
 */
#ifndef VLR_DETMATH_H
#define VLR_DETMATH_H

#if defined(__HIPCC__)
#define VLR_HD __host__ __device__
#else
#define VLR_HD
#endif

/* no fused contraction of a*b+c beyond the explicit fma()s below: results must not depend on whether the target has FMA
 * (gfx950 does, the oracle's x86-64-v2 build does not).  GCC: build with -ffp-contract=off (oracle/Makefile). */
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace vlr_det {

/* exp(x), |x| < 700: x = k ln2 + r, |r| <= ln2/2, degree-14 Taylor polynomial by Horner with fma */
VLR_HD inline double det_exp(double x) {
    if (x != x) return x;
    if (x < -745.0) return 0.0;
    if (x > 709.0) return __builtin_huge_val();
    const double inv_ln2 = 0x1.71547652b82fep+0;
    const double ln2_hi = 0x1.62e42fee00000p-1;   /* upper 32 bits of ln 2 */
    const double ln2_lo = 0x1.a39ef35793c76p-33;  /* ln 2 - ln2_hi */
    double k = __builtin_rint(x * inv_ln2);
    double r = __builtin_fma(-k, ln2_hi, x);
    r = __builtin_fma(-k, ln2_lo, r);
    /* Horner over the Taylor coefficients 1/14! ... 1/0! (nearest doubles) */
    double p = 0x1.93974a8c07c9dp-37;  /* 1/14! */
    p = __builtin_fma(p, r, 0x1.6124613a86d09p-33);  /* 1/13! */
    p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);  /* 1/12! */
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);  /* 1/11! */
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);  /* 1/10! */
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);  /* 1/9! */
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);  /* 1/8! */
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);  /* 1/7! */
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);  /* 1/6! */
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);  /* 1/5! */
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);  /* 1/4! */
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);  /* 1/3! */
    p = __builtin_fma(p, r, 0x1.0000000000000p-1);  /* 1/2! */
    p = __builtin_fma(p, r, 0x1.0000000000000p+0);  /* 1/1! */
    p = __builtin_fma(p, r, 0x1.0000000000000p+0);  /* 1/0! */
    return __builtin_ldexp(p, (int)k);
}

/* ln(1 + s) for s >= 0: u = 1 + s = 2^e m, m in [sqrt(1/2), sqrt 2), ln m = 2 atanh((m-1)/(m+1)) */
VLR_HD inline double det_log1p_pos(double s) {
    if (s != s) return s;
    if (s <= 0.0) return 0.0;
    double u = 1.0 + s;
    int e;
    double m = __builtin_frexp(u, &e);
    if (m < 0x1.6a09e667f3bcdp-1) { m = m * 2.0; e -= 1; }
    double z = (m - 1.0) / (m + 1.0);
    double w = z * z;
    double t = 1.0 / 27.0;
    t = __builtin_fma(t, w, 1.0 / 25.0);
    t = __builtin_fma(t, w, 1.0 / 23.0);
    t = __builtin_fma(t, w, 1.0 / 21.0);
    t = __builtin_fma(t, w, 1.0 / 19.0);
    t = __builtin_fma(t, w, 1.0 / 17.0);
    t = __builtin_fma(t, w, 1.0 / 15.0);
    t = __builtin_fma(t, w, 1.0 / 13.0);
    t = __builtin_fma(t, w, 1.0 / 11.0);
    t = __builtin_fma(t, w, 1.0 / 9.0);
    t = __builtin_fma(t, w, 1.0 / 7.0);
    t = __builtin_fma(t, w, 1.0 / 5.0);
    t = __builtin_fma(t, w, 1.0 / 3.0);
    t = __builtin_fma(t, w, 1.0);
    double lnm = 2.0 * z * t;
    const double ln2_hi = 0x1.62e42fee00000p-1;
    const double ln2_lo = 0x1.a39ef35793c76p-33;
    double de = (double)e;
    return __builtin_fma(de, ln2_hi, __builtin_fma(de, ln2_lo, lnm));
}

/* log2(a) - log2(b) for a, b > 0 (utils/log2_fold_change.rs:17-26).  The l2fc predicates compare this value with literal
 * thresholds, and the VAF bounds inferred from one sample (vaf / 2^value, log2_fold_change.rs:56-93) put the first
 * integration point of the other sample EXACTLY on the threshold whenever 2^value is a power of two: libm's log2 then
 * decides by its last-bit rounding whether that end point counts.  Here the binary exponents are subtracted as
 * integers and only the mantissa parts go through the (deterministic) logarithm, so a ratio of exactly 2^k gives
 * exactly k on every platform. */
VLR_HD inline void det_log2_parts(double x, int* e_out, double* frac_out) {
    int e;
    double m = __builtin_frexp(x, &e);
    if (m < 0x1.6a09e667f3bcdp-1) { m = m * 2.0; e -= 1; }
    double z = (m - 1.0) / (m + 1.0);
    double w = z * z;
    double t = 1.0 / 27.0;
    t = __builtin_fma(t, w, 1.0 / 25.0);
    t = __builtin_fma(t, w, 1.0 / 23.0);
    t = __builtin_fma(t, w, 1.0 / 21.0);
    t = __builtin_fma(t, w, 1.0 / 19.0);
    t = __builtin_fma(t, w, 1.0 / 17.0);
    t = __builtin_fma(t, w, 1.0 / 15.0);
    t = __builtin_fma(t, w, 1.0 / 13.0);
    t = __builtin_fma(t, w, 1.0 / 11.0);
    t = __builtin_fma(t, w, 1.0 / 9.0);
    t = __builtin_fma(t, w, 1.0 / 7.0);
    t = __builtin_fma(t, w, 1.0 / 5.0);
    t = __builtin_fma(t, w, 1.0 / 3.0);
    t = __builtin_fma(t, w, 1.0);
    const double inv_ln2 = 0x1.71547652b82fep+0;
    *e_out = e;
    *frac_out = (2.0 * z * t) * inv_ln2;
}
/* 2^v: exact for integral v, otherwise the deterministic exp of v ln 2 (projection of a VAF through an l2fc
 * predicate, log2_fold_change.rs:58) */
VLR_HD inline double det_exp2(double v) {
    if (!(v > -1000.0 && v < 1000.0)) return det_exp(v * 0x1.62e42fefa39efp-1);
    /* 2^v = 2^k 2^f with k = rint(v), |f| <= 1/2: the rounding of f ln 2 costs half an ulp instead of |v| ulps */
    const double k = __builtin_rint(v), f = v - k;
    if (f == 0.0) return __builtin_ldexp(1.0, (int)k);
    return __builtin_ldexp(det_exp(f * 0x1.62e42fefa39efp-1), (int)k);
}
VLR_HD inline double det_log2_ratio(double a, double b) {
    if (a != a || b != b) return a + b;
    if (a == 0.0 && b == 0.0) return 0.0;
    if (a == 0.0) return -__builtin_huge_val();
    if (b == 0.0) return __builtin_huge_val();
    int ea, eb;
    double fa, fb;
    det_log2_parts(a, &ea, &fa);
    det_log2_parts(b, &eb, &fb);
    return (double)(ea - eb) + (fa - fb);
}

/* Double-double accumulation for the decision sums: sum_i exp(v_i - m) is formed with ~106 significant bits, so the
 * rounded result does not depend on the order of summation (sequential on the CPU, lane-strided plus a butterfly on the
 * GPU) except with probability ~2^-50 per sum. */
struct dd {
    double hi, lo;
};
VLR_HD inline dd dd_two_sum(double a, double b) {
    double s = a + b;
    double bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return dd{s, e};
}
VLR_HD inline dd dd_add(dd a, double x) {
    dd t = dd_two_sum(a.hi, x);
    double lo = t.lo + a.lo;
    double hi = t.hi + lo;
    return dd{hi, lo - (hi - t.hi)};
}
VLR_HD inline dd dd_add_dd(dd a, dd b) {
    dd t = dd_two_sum(a.hi, b.hi);
    double lo = t.lo + (a.lo + b.lo);
    double hi = t.hi + lo;
    return dd{hi, lo - (hi - t.hi)};
}
/* exp(ln_sum_exp(v)) of bio's formula m + ln1p(sum_{i != imax} exp(v_i - m)) given m = max v and the double-double sum
 * over ALL finite terms (the maximum term contributes exactly 1) */
VLR_HD inline double exp_lse_from_sum(double m, dd sum_all) {
    if (m == -__builtin_huge_val()) return 0.0;
    dd rest = dd_add(sum_all, -1.0);
    double s = rest.hi + rest.lo;
    return det_exp(m + det_log1p_pos(s));
}

}  /* namespace vlr_det */
#endif
