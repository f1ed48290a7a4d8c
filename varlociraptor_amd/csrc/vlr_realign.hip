// vlr_realign.hip — gfx950 kernel of the read-vs-allele pair HMM ("next" row #1): the producer of
// prob_alt / prob_ref in `varlociraptor preprocess variants`.
//
// What is computed: bio::stats::pairhmm::PairHMM::prob_related (third-party crate, restated in
// oracle/vlr_realign_oracle.cpp — see its header for what is and is not pinned) over the reference's emission model
//   ReadVsAlleleEmission / ReadEmission        /root/reference/src/variants/evidence/realignment/pairhmm.rs:296-455
//   GapParams, semiglobal start/end            pairhmm.rs:119-205
//   band = edit distance of the hit + EDIT_BAND realignment/mod.rs:519-537, pairhmm.rs:20
// for a batch of (allele window x, read window y) pairs.
//
// How: one wave64 per pair, anti-diagonal wavefront.  Lane l owns read rows 2l and 2l+1 (the reference limits a read window
// to 128 bases, EditDistanceCalculation::max_pattern_len, edit_distance.rs:145-147); at step d a row j works on column
// i = d - j.  The three forward states and the running minimum edit distance of a cell live in registers; a row needs the
// previous step's cell of the row above (wave_shr:1 DPP shift, or its own lane's other register), which one step later is
// its top-left neighbour — no LDS, no barriers.  Arithmetic is linear-space f64 (the reference works in log space): a cell
// costs 3 multiplies + 4 FMAs instead of ~5 exp/log1p; every lane keeps a power-of-two scale for its two rows (rows deep in
// an unrelated read are hundreds of orders of magnitude below the first ones) that is aligned when neighbours exchange cells.
// No MFMA (a recurrence, not a contraction); HBM traffic is the two sequences and the qualities, a few hundred bytes per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/vlr.h"

namespace vlr {

struct RealignArgs {
    int64_t n_pairs;
    const uint32_t* x_offset;
    const uint8_t* x_bases;
    const uint32_t* y_offset;
    const uint8_t* y_bases;
    const uint8_t* y_quals;
    const int32_t* max_edit_dist;
    double pn, pnx, pny, pgx, pgy, pgxe, pgye;  // linear: P(no gap), P(leave x-gap), P(leave y-gap), gap opens, extends
    double* ln_prob;
};

__device__ __forceinline__ int up(int b) { return (b >= 'a' && b <= 'z') ? b - 32 : b; }

// value of lane l-1 (lane 0: zero — bound_ctrl supplies it, no copy of an edge value into the destination first)
__device__ __forceinline__ double shr1z(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// value of lane l-1 (lane 0: `edge`)
__device__ __forceinline__ unsigned shr1(unsigned v, unsigned edge) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false);
}

constexpr unsigned kBig = 0x3fffffffu;  // "unreachable" edit distance (adding one cannot wrap)

__device__ __forceinline__ void realign_one(const RealignArgs& a, const int64_t pair, const int lane) {
    const uint32_t x0 = a.x_offset[pair], y0 = a.y_offset[pair];
    const int len_x = (int)(a.x_offset[pair + 1] - x0), len_y = (int)(a.y_offset[pair + 1] - y0);
    const int med_max = a.max_edit_dist ? a.max_edit_dist[pair] : -1;
    const bool banded = med_max >= 0;
    if (len_y > 128 || len_y <= 0 || len_x <= 0) {
        if (lane == 0) a.ln_prob[pair] = (len_x <= 0 || len_y <= 0) ? -__builtin_huge_val() : __builtin_nan("");
        return;
    }
    // per-row emission constants (ReadEmission::new, pairhmm.rs:406-428; PROB_CONFUSION pairhmm.rs:22-24).  Rows beyond the
    // read get zero emissions: every state of such a row stays exactly zero without a select in the loop.
    int yb[2];
    double e_match[2], e_mis[2], e_ins[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * lane + r;
        const bool rowon = j < len_y;
        const int q = rowon ? a.y_quals[y0 + j] : 0;
        yb[r] = rowon ? up(a.y_bases[y0 + j]) : 0;
        const double mis = exp(-(double)q * 2.302585092994046 / 10.0);  // P(miscall) = 10^(-q/10)
        e_match[r] = rowon ? 1.0 - mis : 0.0;
        e_mis[r] = rowon ? mis * 0.3333 : 0.0;
        e_ins[r] = rowon ? mis : 0.0;
    }
    // state of the cell each row computed at the previous step (its "left" neighbour now) ...
    double M1[2] = {0.0, 0.0}, X1[2] = {0.0, 0.0}, Y1[2] = {0.0, 0.0};
    unsigned E1[2] = {kBig, kBig};
    // ... and of the row above one step earlier (the "top-left" neighbour now)
    double Mt[2] = {0.0, 0.0}, Xt[2] = {0.0, 0.0}, Yt[2] = {0.0, 0.0};
    unsigned Et[2] = {kBig, kBig};
    double total = 0.0;  // sum over columns of the last row's three states (free end gap in x)
    int scale = 0;       // this lane's states and total carry a factor 2^scale
    const int last_row = len_y - 1;
    const int lr = last_row & 1;             // which of its two rows the owner lane sums (uniform)
    const bool owner_lane = lane == (last_row >> 1);
    const int nsteps = len_x + len_y - 1;
    // x bases travel with the wavefront: row 2l works on column d - 2l, row 2l+1 on the column row 2l had one step earlier, and
    // row 2l's column is the one row 2(l-1)+1 had one step earlier.  So every base is loaded once (64 at a time, lane-
    // contiguous), enters at lane 0 and moves down the lanes by one DPP shift per step; columns outside the allele carry 0.
    int xchunk = 0, xb0 = 0, xb1 = 0;
    for (int d = 0; d < nsteps; ++d) {
        if ((d & 63) == 0) {
            const int i = d + lane;
            xchunk = (i < len_x) ? up(a.x_bases[x0 + i]) : 0;
        }
        const int xnew = __builtin_amdgcn_readlane(xchunk, d & 63);
        const int prev1 = xb1;
        xb1 = xb0;
        xb0 = (int)shr1((unsigned)prev1 /* lane l-1's row-1 base of the previous step */, (unsigned)xnew);
        // top neighbour = previous step's cell of row j-1.  Row -1 is the virtual start row: as "top" (same column) it is
        // empty, as "top-left" (previous column) it carries the free start mass one with edit distance zero.
        double Mu[2], Xu[2], Yu[2];
        unsigned Eu[2];
        Mu[0] = shr1z(M1[1]); Xu[0] = shr1z(X1[1]); Yu[0] = shr1z(Y1[1]); Eu[0] = shr1(E1[1], kBig);
        {
            // the lane above stores its states with its own power-of-two scale: bring them to this lane's.  A lane that holds
            // nothing yet (rows not reached, or everything outside the band) simply adopts the scale of the lane above.
            // Skipped altogether while all lanes agree (the common case: scales only move at the checks below).
            const int nb = (int)shr1((unsigned)scale, (unsigned)scale);
            if (__ballot(scale != nb)) {
                const double mass = ((M1[0] + M1[1]) + (X1[0] + X1[1])) + ((Y1[0] + Y1[1]) + (Mt[0] + Mt[1])) + ((Xt[0] + Xt[1]) + (Yt[0] + Yt[1])) + total;
                if (mass == 0.0) scale = nb;
                int dsc = scale - nb;
                dsc = dsc > 1000 ? 1000 : dsc < -1000 ? -1000 : dsc;
                const double f = __builtin_ldexp(1.0, dsc);
                Mu[0] *= f; Xu[0] *= f; Yu[0] *= f;
            }
        }
        Mu[1] = M1[0]; Xu[1] = X1[0]; Yu[1] = Y1[0]; Eu[1] = E1[0];
        // virtual start row: the top-left neighbour of row 0 holds mass one in every column (free start gap in x)
        if (lane == 0) { Mt[0] = __builtin_ldexp(1.0, scale); Et[0] = 0u; }
        double Mn[2], Xn[2], Yn[2];
        unsigned En[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = d - (2 * lane + r);
            const bool incol = (unsigned)i < (unsigned)len_x;
            const int xb = r == 0 ? xb0 : xb1;
            const bool is_match = xb == yb[r];
            // top-left (j-1, i-1), left (j, i-1), top (j-1, i): cells outside the matrix are exactly zero by construction (rows
            // start from zero states, the row above is zero before its first column)
            const double emit = is_match ? e_match[r] : e_mis[r];
            const double m = emit * (a.pn * Mt[r] + a.pny * Xt[r] + a.pnx * Yt[r]);
            const double x = a.pgy * M1[r] + a.pgye * X1[r];                  // gap in y: x_i alone (prob_emit_x = 1)
            const double y = e_ins[r] * (a.pgx * Mu[r] + a.pgxe * Yu[r]);    // gap in x: y_j alone
            bool live = incol;
            unsigned e = kBig;
            if (banded) {  // (uniform per pair)
                const unsigned etl = Et[r], eu = Eu[r], el = E1[r];
                const unsigned emin = min(etl, min(eu, el));
                live = incol && !(emin > (unsigned)med_max);
                e = live ? min(min(is_match ? etl : etl + 1u, min(eu + 1u, el + 1u)), kBig) : kBig;
            }
            // beyond the last column (and outside the band) a row keeps nothing
            Mn[r] = live ? m : 0.0; Xn[r] = live ? x : 0.0; Yn[r] = live ? y : 0.0;
            En[r] = e;
        }
        {
            const double s = (Mn[lr] + Xn[lr]) + Yn[lr];
            total += owner_lane ? s : 0.0;  // (zero outside the columns of the allele)
        }
        // this step's "top" of a row is its "top-left" at the next step
#pragma unroll
        for (int r = 0; r < 2; ++r) { Mt[r] = Mu[r]; Xt[r] = Xu[r]; Yt[r] = Yu[r]; Et[r] = Eu[r]; M1[r] = Mn[r]; X1[r] = Xn[r]; Y1[r] = Yn[r]; E1[r] = En[r]; }
        // underflow guard, every 8 steps (a Q93 mismatch shrinks a state by 1e-10: 1e-80 between two checks).  Rows deep in
        // the read carry far smaller numbers than the first rows, so every LANE keeps its own power-of-two scale: when its
        // largest state has dropped below 2^-200 it is brought back to ~1 (exact), unless what the lane has collected for
        // the result already outweighs anything its states can still add.
        if ((d & 7) == 7) {
            double mx = fmax(fmax(fmax(M1[0], X1[0]), fmax(Y1[0], M1[1])), fmax(X1[1], Y1[1]));
            mx = fmax(mx, fmax(fmax(Mt[0], Xt[0]), fmax(fmax(Yt[0], Mt[1]), fmax(Xt[1], Yt[1]))));
            int ex = 0;
            (void)__builtin_frexp(mx, &ex);
            // (scaled DOWN as well: the mass of a lane grows again when the wavefront reaches the columns the read aligns to)
            const bool resc = mx > 0.0 && (ex > 200 || (ex < -200 && !(total > mx * 0x1p60)));
            if (__ballot(resc)) {
                const int sh0 = -ex > 1000 ? 1000 : -ex < -1000 ? -1000 : -ex;
                const int sh = resc ? sh0 : 0;
                const double f1 = __builtin_ldexp(1.0, sh);
#pragma unroll
                for (int r = 0; r < 2; ++r) { M1[r] *= f1; X1[r] *= f1; Y1[r] *= f1; Mt[r] *= f1; Xt[r] *= f1; Yt[r] *= f1; }
                total *= f1;
                scale += sh;
            }
        }
    }
    // the lane that owns the last row holds the sum
    const int owner = last_row >> 1;
    total = __shfl(total, owner);
    scale = __shfl(scale, owner);
    if (lane == 0) {
        double p = (total > 0.0) ? log(total) - (double)scale * 0.6931471805599453 : -__builtin_huge_val();
        a.ln_prob[pair] = p > 0.0 ? 0.0 : p;  // "sum of paths can exceed probability 1.0"
    }
}

__global__ void __launch_bounds__(64) vlr_realign_kernel(RealignArgs a) {
    const int64_t pair = blockIdx.x;
    if (pair >= a.n_pairs) return;
    realign_one(a, pair, threadIdx.x);
}

// ---- two pairs per wave ------------------------------------------------------------------------------------------------
// A read window of at most 64 bases occupies 32 lanes of the wavefront above.  Workgroup w takes the pairs 2w and 2w + 1 — the
// reference and the alt allele of one read are adjacent in a batch and share the read window — and, when both windows are
// short, runs them side by side: lanes 0-31 pair 2w, lanes 32-63 pair 2w + 1.  Everything that was wave-uniform per pair
// (lengths, band, owner lane, step count) is per half; the wave_shr:1 shifts cross the half boundary, so lane 32 takes the edge
// value instead of lane 31's.  Same arithmetic per cell in the same order: results are bit-identical to the one-pair kernel.
__global__ void __launch_bounds__(64) vlr_realign_kernel2(RealignArgs a) {
    const int64_t pair0 = 2 * (int64_t)blockIdx.x;
    if (pair0 >= a.n_pairs) return;
    const int lane = threadIdx.x;
    const bool have2 = pair0 + 1 < a.n_pairs;
    const int ly0 = (int)(a.y_offset[pair0 + 1] - a.y_offset[pair0]);
    const int ly1 = have2 ? (int)(a.y_offset[pair0 + 2] - a.y_offset[pair0 + 1]) : 0;
    const int lx0 = (int)(a.x_offset[pair0 + 1] - a.x_offset[pair0]);
    const int lx1 = have2 ? (int)(a.x_offset[pair0 + 2] - a.x_offset[pair0 + 1]) : 0;
    if (!(have2 && ly0 > 0 && ly1 > 0 && ly0 <= 64 && ly1 <= 64 && lx0 > 0 && lx1 > 0)) {  // (uniform) one after the other
        realign_one(a, pair0, lane);
        if (have2) realign_one(a, pair0 + 1, lane);
        return;
    }
    const int half = lane >> 5, hl = lane & 31;
    const bool edge = hl == 0;  // lanes 0 and 32: row 0 of their pair
    const int64_t pair = pair0 + half;
    const uint32_t x0 = a.x_offset[pair], y0 = a.y_offset[pair];
    const int len_x = half ? lx1 : lx0, len_y = half ? ly1 : ly0;
    const int med_max = a.max_edit_dist ? a.max_edit_dist[pair] : -1;
    const bool banded = med_max >= 0;
    int yb[2];
    double e_match[2], e_mis[2], e_ins[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * hl + r;
        const bool rowon = j < len_y;
        const int q = rowon ? a.y_quals[y0 + j] : 0;
        yb[r] = rowon ? up(a.y_bases[y0 + j]) : 0;
        const double mis = exp(-(double)q * 2.302585092994046 / 10.0);
        e_match[r] = rowon ? 1.0 - mis : 0.0;
        e_mis[r] = rowon ? mis * 0.3333 : 0.0;
        e_ins[r] = rowon ? mis : 0.0;
    }
    double M1[2] = {0.0, 0.0}, X1[2] = {0.0, 0.0}, Y1[2] = {0.0, 0.0};
    unsigned E1[2] = {kBig, kBig};
    double Mt[2] = {0.0, 0.0}, Xt[2] = {0.0, 0.0}, Yt[2] = {0.0, 0.0};
    unsigned Et[2] = {kBig, kBig};
    double total = 0.0;
    int scale = 0;
    const int last_row = len_y - 1;
    const int lr = last_row & 1;
    const bool owner_lane = hl == (last_row >> 1);
    const int nsteps = len_x + len_y - 1;
    const int ns0 = __builtin_amdgcn_readlane(nsteps, 0), ns1 = __builtin_amdgcn_readlane(nsteps, 32);
    const int nmax = ns0 > ns1 ? ns0 : ns1;
    int xchunk = 0, xb0 = 0, xb1 = 0;
    for (int d = 0; d < nmax; ++d) {
        if ((d & 31) == 0) {
            const int i = d + hl;
            xchunk = (i < len_x) ? up(a.x_bases[x0 + i]) : 0;
        }
        const int xn0 = __builtin_amdgcn_readlane(xchunk, d & 31), xn1 = __builtin_amdgcn_readlane(xchunk, 32 + (d & 31));
        const int xnew = half ? xn1 : xn0;
        const int prev1 = xb1;
        xb1 = xb0;
        {
            const int sh = (int)shr1((unsigned)prev1, (unsigned)xnew);
            xb0 = edge ? xnew : sh;
        }
        double Mu[2], Xu[2], Yu[2];
        unsigned Eu[2];
        {
            const double m = shr1z(M1[1]), x = shr1z(X1[1]), y = shr1z(Y1[1]);
            const unsigned e = shr1(E1[1], kBig);
            Mu[0] = edge ? 0.0 : m; Xu[0] = edge ? 0.0 : x; Yu[0] = edge ? 0.0 : y; Eu[0] = edge ? kBig : e;
        }
        {
            const int nbs = (int)shr1((unsigned)scale, (unsigned)scale);
            const int nb = edge ? scale : nbs;
            if (__ballot(scale != nb)) {
                const double mass = ((M1[0] + M1[1]) + (X1[0] + X1[1])) + ((Y1[0] + Y1[1]) + (Mt[0] + Mt[1])) + ((Xt[0] + Xt[1]) + (Yt[0] + Yt[1])) + total;
                if (mass == 0.0) scale = nb;
                int dsc = scale - nb;
                dsc = dsc > 1000 ? 1000 : dsc < -1000 ? -1000 : dsc;
                const double f = __builtin_ldexp(1.0, dsc);
                Mu[0] *= f; Xu[0] *= f; Yu[0] *= f;
            }
        }
        Mu[1] = M1[0]; Xu[1] = X1[0]; Yu[1] = Y1[0]; Eu[1] = E1[0];
        if (edge) { Mt[0] = __builtin_ldexp(1.0, scale); Et[0] = 0u; }
        double Mn[2], Xn[2], Yn[2];
        unsigned En[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = d - (2 * hl + r);
            const bool incol = (unsigned)i < (unsigned)len_x;
            const int xb = r == 0 ? xb0 : xb1;
            const bool is_match = xb == yb[r];
            const double emit = is_match ? e_match[r] : e_mis[r];
            const double m = emit * (a.pn * Mt[r] + a.pny * Xt[r] + a.pnx * Yt[r]);
            const double x = a.pgy * M1[r] + a.pgye * X1[r];
            const double y = e_ins[r] * (a.pgx * Mu[r] + a.pgxe * Yu[r]);
            bool live = incol;
            unsigned e = kBig;
            {
                const unsigned etl = Et[r], eu = Eu[r], el = E1[r];
                const unsigned emin = min(etl, min(eu, el));
                const bool in_band = !(emin > (unsigned)med_max);
                live = incol && (!banded || in_band);
                const unsigned eb = live ? min(min(is_match ? etl : etl + 1u, min(eu + 1u, el + 1u)), kBig) : kBig;
                e = banded ? eb : kBig;
            }
            Mn[r] = live ? m : 0.0; Xn[r] = live ? x : 0.0; Yn[r] = live ? y : 0.0;
            En[r] = e;
        }
        {
            const double s = (Mn[lr] + Xn[lr]) + Yn[lr];
            total += owner_lane ? s : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) { Mt[r] = Mu[r]; Xt[r] = Xu[r]; Yt[r] = Yu[r]; Et[r] = Eu[r]; M1[r] = Mn[r]; X1[r] = Xn[r]; Y1[r] = Yn[r]; E1[r] = En[r]; }
        if ((d & 7) == 7) {
            double mx = fmax(fmax(fmax(M1[0], X1[0]), fmax(Y1[0], M1[1])), fmax(X1[1], Y1[1]));
            mx = fmax(mx, fmax(fmax(Mt[0], Xt[0]), fmax(fmax(Yt[0], Mt[1]), fmax(Xt[1], Yt[1]))));
            int ex = 0;
            (void)__builtin_frexp(mx, &ex);
            const bool resc = mx > 0.0 && (ex > 200 || (ex < -200 && !(total > mx * 0x1p60)));
            if (__ballot(resc)) {
                const int sh0 = -ex > 1000 ? 1000 : -ex < -1000 ? -1000 : -ex;
                const int sh = resc ? sh0 : 0;
                const double f1 = __builtin_ldexp(1.0, sh);
#pragma unroll
                for (int r = 0; r < 2; ++r) { M1[r] *= f1; X1[r] *= f1; Y1[r] *= f1; Mt[r] *= f1; Xt[r] *= f1; Yt[r] *= f1; }
                total *= f1;
                scale += sh;
            }
        }
    }
    const int owner = 32 * half + (last_row >> 1);
    total = __shfl(total, owner);
    scale = __shfl(scale, owner);
    if (edge) {
        double p = (total > 0.0) ? log(total) - (double)scale * 0.6931471805599453 : -__builtin_huge_val();
        a.ln_prob[pair] = p > 0.0 ? 0.0 : p;
    }
}

// ---- `homopolymer` realignment mode --------------------------------------------------------------------------------------
// HomopolyPairHMMRealigner::calculate_prob_allele (realignment/mod.rs:680-730): bio's HomopolyPairHMM::prob_related with the
// reference's HopParams (pairhmm.rs:207-295) — the pair HMM above plus, per base, hop states that emit one more copy of the
// homopolymer base in the read (HopX_b: y_j == b alone, emitted like a matching base — oracle/vlr_realign_oracle.cpp says on which
// evidence) or in the allele (HopY_b: x_i == b alone), entered from the match state of
// the same base and left to a match state only.  Restated in oracle/vlr_realign_oracle.cpp (vlro_homopoly_prob_related, PARITY
// UNPINNED: the recursion lives in the un-vendored crate bio) with all fourteen states spelled out; here the model is folded:
// the match state of a cell is the one of its allele base x_i, so of the four HopX_b / HopY_b / Match_b at most one each is
// non-zero per cell — HopX needs y_j == x_i, HopY needs x_i == x_{i-1} — and a cell carries five values {M, X, Y, P, Q}
// (X = x_i alone after a gap open, Y = y_j alone, P = HopY, Q = HopX).  The transition out of a match or hop state depends on the
// base of the PREVIOUS column (1 - (gap_x + gap_y + hop_x(b') + hop_y(b')), 1 - hop_extend(b')): the per-base constants sit in a
// 5 x 8 table in LDS (row 4: any other base, no hops), row 0 of a lane looks up the base entering its column, row 1 inherits
// what row 0 held one step earlier.  Same wavefront, scaling and band as realign_one.
struct HomopolyArgs {
    RealignArgs r;
    double hx[4], hy[4], hxe[4], hye[4];  // linear: start / extend a homopolymer run in the read (x gap) / in the allele (y gap)
};
__device__ __forceinline__ int base_index(int b) { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : 4; }

__global__ void __launch_bounds__(64) vlr_homopoly_kernel(HomopolyArgs h) {
    const RealignArgs& a = h.r;
    const int64_t pair = blockIdx.x;
    if (pair >= a.n_pairs) return;
    const int lane = threadIdx.x;
    __shared__ double tab[5][8];  // {hop_x, hop_y, hop_x_extend, hop_y_extend, match->match, leave hop x, leave hop y, -}
    if (lane < 5) {
        const bool acgt = lane < 4;
        const int b = acgt ? lane : 0;
        const double hx = acgt ? h.hx[b] : 0.0, hy = acgt ? h.hy[b] : 0.0, hxe = acgt ? h.hxe[b] : 0.0, hye = acgt ? h.hye[b] : 0.0;
        tab[lane][0] = hx; tab[lane][1] = hy; tab[lane][2] = hxe; tab[lane][3] = hye;
        tab[lane][4] = 1.0 - (((a.pgx + a.pgy) + hx) + hy);
        tab[lane][5] = 1.0 - hxe; tab[lane][6] = 1.0 - hye; tab[lane][7] = 0.0;
    }
    __syncthreads();
    const uint32_t x0 = a.x_offset[pair], y0 = a.y_offset[pair];
    const int len_x = (int)(a.x_offset[pair + 1] - x0), len_y = (int)(a.y_offset[pair + 1] - y0);
    const int med_max = a.max_edit_dist ? a.max_edit_dist[pair] : -1;
    const bool banded = med_max >= 0;
    if (len_y > 128 || len_y <= 0 || len_x <= 0) {
        if (lane == 0) a.ln_prob[pair] = (len_x <= 0 || len_y <= 0) ? -__builtin_huge_val() : __builtin_nan("");
        return;
    }
    int yb[2];
    double e_match[2], e_mis[2], e_ins[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * lane + r;
        const bool rowon = j < len_y;
        const int q = rowon ? a.y_quals[y0 + j] : 0;
        yb[r] = rowon ? up(a.y_bases[y0 + j]) : 0;
        const double mis = exp(-(double)q * 2.302585092994046 / 10.0);
        e_match[r] = rowon ? 1.0 - mis : 0.0;
        e_mis[r] = rowon ? mis * 0.3333 : 0.0;
        e_ins[r] = rowon ? mis : 0.0;
    }
    double M1[2] = {0.0, 0.0}, X1[2] = {0.0, 0.0}, Y1[2] = {0.0, 0.0}, P1[2] = {0.0, 0.0}, Q1[2] = {0.0, 0.0};
    unsigned E1[2] = {kBig, kBig};
    double Mt[2] = {0.0, 0.0}, Xt[2] = {0.0, 0.0}, Yt[2] = {0.0, 0.0}, Pt[2] = {0.0, 0.0}, Qt[2] = {0.0, 0.0};
    unsigned Et[2] = {kBig, kBig};
    // per-base constants of the column each row works on (c*) and of the column before it (p*: what the top-left cell leaves with)
    double chx[2] = {0.0, 0.0}, chy[2] = {0.0, 0.0}, chxe[2] = {0.0, 0.0}, chye[2] = {0.0, 0.0};
    double ctm[2] = {a.pn, a.pn}, clx[2] = {1.0, 1.0}, cly[2] = {1.0, 1.0};
    double ptm[2] = {a.pn, a.pn}, plx[2] = {1.0, 1.0}, ply[2] = {1.0, 1.0};
    double total = 0.0;
    int scale = 0;
    const int last_row = len_y - 1;
    const int lr = last_row & 1;
    const bool owner_lane = lane == (last_row >> 1);
    const int nsteps = len_x + len_y - 1;
    int xchunk = 0, xb0 = 0, xb1 = 0, xp0 = 0, xp1 = 0;
    for (int d = 0; d < nsteps; ++d) {
        if ((d & 63) == 0) {
            const int i = d + lane;
            xchunk = (i < len_x) ? up(a.x_bases[x0 + i]) : 0;
        }
        const int xnew = __builtin_amdgcn_readlane(xchunk, d & 63);
        xp1 = xb1; xp0 = xb0;  // the bases of the previous column of each row
        const int prev1 = xb1;
        xb1 = xb0;
        xb0 = (int)shr1((unsigned)prev1, (unsigned)xnew);
        // constants: row 1 inherits row 0's of the previous step (same column), row 0 looks its new base up
        ptm[1] = ctm[1]; plx[1] = clx[1]; ply[1] = cly[1];
        chx[1] = chx[0]; chy[1] = chy[0]; chxe[1] = chxe[0]; chye[1] = chye[0]; ctm[1] = ctm[0]; clx[1] = clx[0]; cly[1] = cly[0];
        ptm[0] = ctm[0]; plx[0] = clx[0]; ply[0] = cly[0];
        {
            const double* t = tab[base_index(xb0)];
            chx[0] = t[0]; chy[0] = t[1]; chxe[0] = t[2]; chye[0] = t[3]; ctm[0] = t[4]; clx[0] = t[5]; cly[0] = t[6];
        }
        double Mu[2], Xu[2], Yu[2], Qu[2];
        unsigned Eu[2];
        Mu[0] = shr1z(M1[1]); Xu[0] = shr1z(X1[1]); Yu[0] = shr1z(Y1[1]); Eu[0] = shr1(E1[1], kBig);
        double Pu0 = shr1z(P1[1]);
        Qu[0] = shr1z(Q1[1]);
        {
            const int nb = (int)shr1((unsigned)scale, (unsigned)scale);
            if (__ballot(scale != nb)) {
                const double mass = ((M1[0] + M1[1]) + (X1[0] + X1[1])) + ((Y1[0] + Y1[1]) + (Mt[0] + Mt[1])) + ((Xt[0] + Xt[1]) + (Yt[0] + Yt[1])) +
                                    ((P1[0] + P1[1]) + (Q1[0] + Q1[1])) + ((Pt[0] + Pt[1]) + (Qt[0] + Qt[1])) + total;
                if (mass == 0.0) scale = nb;
                int dsc = scale - nb;
                dsc = dsc > 1000 ? 1000 : dsc < -1000 ? -1000 : dsc;
                const double f = __builtin_ldexp(1.0, dsc);
                Mu[0] *= f; Xu[0] *= f; Yu[0] *= f; Pu0 *= f; Qu[0] *= f;
            }
        }
        Mu[1] = M1[0]; Xu[1] = X1[0]; Yu[1] = Y1[0]; Qu[1] = Q1[0]; Eu[1] = E1[0];
        const double Pu1 = P1[0];
        // virtual start row: the top-left neighbour of row 0 holds mass one in every column, left with 1 - (gap_x + gap_y)
        double tm0 = ptm[0];
        if (lane == 0) { Mt[0] = __builtin_ldexp(1.0, scale); Et[0] = 0u; tm0 = a.pn; }
        double Mn[2], Xn[2], Yn[2], Pn[2], Qn[2];
        unsigned En[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = d - (2 * lane + r);
            const bool incol = (unsigned)i < (unsigned)len_x;
            const int xb = r == 0 ? xb0 : xb1, xp = r == 0 ? xp0 : xp1;
            const bool is_match = xb == yb[r];
            const double emit = is_match ? e_match[r] : e_mis[r];
            const double tm = r == 0 ? tm0 : ptm[1];
            const double m = emit * ((tm * Mt[r] + a.pny * Xt[r] + a.pnx * Yt[r]) + (ply[r] * Pt[r] + plx[r] * Qt[r]));
            const double x = a.pgy * M1[r] + a.pgye * X1[r];
            const double y = e_ins[r] * (a.pgx * Mu[r] + a.pgxe * Yu[r]);
            const double pp = (xb == xp && i > 0) ? chy[r] * M1[r] + chye[r] * P1[r] : 0.0;           // x_i == x_{i-1} alone
            const double qq = is_match ? e_match[r] * (chx[r] * Mu[r] + chxe[r] * Qu[r]) : 0.0;        // y_j == x_i alone, emitted like a matching base
            bool live = incol;
            unsigned e = kBig;
            if (banded) {
                const unsigned etl = Et[r], eu = Eu[r], el = E1[r];
                const unsigned emin = min(etl, min(eu, el));
                live = incol && !(emin > (unsigned)med_max);
                e = live ? min(min(is_match ? etl : etl + 1u, min(eu + 1u, el + 1u)), kBig) : kBig;
            }
            Mn[r] = live ? m : 0.0; Xn[r] = live ? x : 0.0; Yn[r] = live ? y : 0.0; Pn[r] = live ? pp : 0.0; Qn[r] = live ? qq : 0.0;
            En[r] = e;
        }
        {
            const double s = ((Mn[lr] + Xn[lr]) + Yn[lr]) + (Pn[lr] + Qn[lr]);
            total += owner_lane ? s : 0.0;
        }
        Mt[0] = Mu[0]; Xt[0] = Xu[0]; Yt[0] = Yu[0]; Pt[0] = Pu0; Qt[0] = Qu[0]; Et[0] = Eu[0];
        Mt[1] = Mu[1]; Xt[1] = Xu[1]; Yt[1] = Yu[1]; Pt[1] = Pu1; Qt[1] = Qu[1]; Et[1] = Eu[1];
#pragma unroll
        for (int r = 0; r < 2; ++r) { M1[r] = Mn[r]; X1[r] = Xn[r]; Y1[r] = Yn[r]; P1[r] = Pn[r]; Q1[r] = Qn[r]; E1[r] = En[r]; }
        if ((d & 7) == 7) {
            double mx = fmax(fmax(fmax(M1[0], X1[0]), fmax(Y1[0], M1[1])), fmax(X1[1], Y1[1]));
            mx = fmax(mx, fmax(fmax(Mt[0], Xt[0]), fmax(fmax(Yt[0], Mt[1]), fmax(Xt[1], Yt[1]))));
            mx = fmax(mx, fmax(fmax(fmax(P1[0], P1[1]), fmax(Q1[0], Q1[1])), fmax(fmax(Pt[0], Pt[1]), fmax(Qt[0], Qt[1]))));
            int ex = 0;
            (void)__builtin_frexp(mx, &ex);
            const bool resc = mx > 0.0 && (ex > 200 || (ex < -200 && !(total > mx * 0x1p60)));
            if (__ballot(resc)) {
                const int sh0 = -ex > 1000 ? 1000 : -ex < -1000 ? -1000 : -ex;
                const int sh = resc ? sh0 : 0;
                const double f1 = __builtin_ldexp(1.0, sh);
#pragma unroll
                for (int r = 0; r < 2; ++r) { M1[r] *= f1; X1[r] *= f1; Y1[r] *= f1; P1[r] *= f1; Q1[r] *= f1; Mt[r] *= f1; Xt[r] *= f1; Yt[r] *= f1; Pt[r] *= f1; Qt[r] *= f1; }
                total *= f1;
                scale += sh;
            }
        }
    }
    const int owner = last_row >> 1;
    total = __shfl(total, owner);
    scale = __shfl(scale, owner);
    if (lane == 0) {
        double p = (total > 0.0) ? log(total) - (double)scale * 0.6931471805599453 : -__builtin_huge_val();
        a.ln_prob[pair] = p > 0.0 ? 0.0 : p;
    }
}

// ---- edit-distance pre-filter ----------------------------------------------------------------------------------------
// EditDistanceCalculation::calc_best_hit (edit_distance.rs:164-260; bio Myers `find_all_lazy`): the smallest semiglobal edit
// distance of the read window against the allele window (free start and end in the allele), the first allele position at
// which an alignment with that distance ends, and the number of such end positions.  The pair HMM is banded to
// distance + EDIT_BAND (realignment/mod.rs:519-537).  Same wavefront as the pair HMM, integers only: lane l owns rows 2l and
// 2l+1, a cell is min(top-left + mismatch, top + 1, left + 1); row -1 is all zero (free start), column -1 of row j is j + 1.
struct EditArgs {
    int64_t n_pairs;
    const uint32_t* x_offset;
    const uint8_t* x_bases;
    const uint32_t* y_offset;
    const uint8_t* y_bases;
    int32_t* dist;
    int32_t* end;
    int32_t* n_hits;
};

__global__ void __launch_bounds__(64) vlr_edit_kernel(EditArgs a) {
    const int64_t pair = blockIdx.x;
    if (pair >= a.n_pairs) return;
    const int lane = threadIdx.x;
    const uint32_t x0 = a.x_offset[pair], y0 = a.y_offset[pair];
    const int len_x = (int)(a.x_offset[pair + 1] - x0), len_y = (int)(a.y_offset[pair + 1] - y0);
    if (len_y > 128 || len_y <= 0 || len_x <= 0) {
        if (lane == 0) { a.dist[pair] = -1; if (a.end) a.end[pair] = -1; if (a.n_hits) a.n_hits[pair] = 0; }
        return;
    }
    int yb[2];
    unsigned E1[2], Et[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * lane + r;
        yb[r] = j < len_y ? up(a.y_bases[y0 + j]) : 0;
        E1[r] = (unsigned)(j + 1);  // the cell left of column 0 of row j
        Et[r] = (unsigned)j;        // (j-1, -1)
    }
    const int last_row = len_y - 1;
    const int lr = last_row & 1;
    const bool owner_lane = lane == (last_row >> 1);
    unsigned best = kBig;
    int best_end = 0, nbest = 0;
    const int nsteps = len_x + len_y - 1;
    int xchunk = 0, xb0 = 0, xb1 = 0;
    for (int d = 0; d < nsteps; ++d) {
        if ((d & 63) == 0) {
            const int i = d + lane;
            xchunk = (i < len_x) ? up(a.x_bases[x0 + i]) : 0;
        }
        const int xnew = __builtin_amdgcn_readlane(xchunk, d & 63);
        const int prev1 = xb1;
        xb1 = xb0;
        xb0 = (int)shr1((unsigned)prev1, (unsigned)xnew);
        unsigned Eu[2];
        Eu[0] = shr1(E1[1], 0u);  // row -1 for lane 0: zero in every column
        Eu[1] = E1[0];
        if (lane == 0) Et[0] = 0u;
        unsigned En[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = d - (2 * lane + r);
            const bool incol = (unsigned)i < (unsigned)len_x;
            const int xb = r == 0 ? xb0 : xb1;
            const unsigned e = min(Et[r] + (xb == yb[r] ? 0u : 1u), min(Eu[r], E1[r]) + 1u);
            En[r] = incol ? e : E1[r];  // before its first column a row keeps the value left of column 0
        }
        {
            const int i = d - last_row;
            const bool hit = owner_lane && (unsigned)i < (unsigned)len_x;
            const unsigned e = En[lr];
            if (hit && e < best) { best = e; best_end = i + 1; nbest = 0; }
            if (hit && e == best) nbest += 1;
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) { Et[r] = Eu[r]; E1[r] = En[r]; }
    }
    if (owner_lane) {
        a.dist[pair] = (int32_t)best;
        if (a.end) a.end[pair] = best_end;
        if (a.n_hits) a.n_hits[pair] = nbest;
    }
}

// ---- `fast` realignment mode -------------------------------------------------------------------------------------------
// PathHMMRealigner::calculate_prob_allele (realignment/mod.rs:547-678): instead of summing over all alignments, the path
// probability of the optimal edit-distance alignments of the best hits (bio Myers traceback, edit_distance.rs:164-260) — transition
// terms by the previous operation (no_gap / close_gap / gap open / extend-or-reopen, mod.rs:560-584), emissions of the
// ReadVsAlleleEmission model — and the best of them.  Which of several co-optimal alignments the crate's traceback returns is
// not specified by the reference; this kernel takes the BEST path probability over ALL alignments of minimal semiglobal edit
// distance (an upper bound of, and with a unique optimal alignment equal to, the reference's value).
// Same wavefront as the pair HMM, in max-plus form over (edit distance, ln probability) pairs ordered lexicographically
// (smaller distance first, then larger probability) per state {match, deletion, insertion}: adds and compares only, log space,
// no scaling.  Cell (j, i) = read bases 0..j and allele bases ..i consumed; row -1 is the free start in every column.
struct PathArgs {
    int64_t n_pairs;
    const uint32_t* x_offset;
    const uint8_t* x_bases;
    const uint32_t* y_offset;
    const uint8_t* y_bases;
    const uint8_t* y_quals;
    double no_gap, close_x, close_y, gap_x, gap_y, reopen_x, reopen_y;  // ln; reopen_* = ln(extend + close * open) (mod.rs:576-584)
    double* ln_prob;
};
struct DP { unsigned d; double p; };
__device__ __forceinline__ DP dp_best(DP a, DP b) {
    const bool ta = (a.d < b.d) | ((a.d == b.d) & (a.p > b.p));
    DP r; r.d = ta ? a.d : b.d; r.p = ta ? a.p : b.p; return r;
}
__device__ __forceinline__ DP dp_step(DP a, unsigned cost, double lp) {  // unreachable stays unreachable
    DP r; r.d = a.d >= kBig ? kBig : a.d + cost; r.p = a.d >= kBig ? -__builtin_huge_val() : a.p + lp; return r;
}
__device__ __forceinline__ double shr1d(double v, double edge) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(edge), lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ DP dp_shr1(DP v) { DP r; r.d = shr1(v.d, kBig); r.p = shr1d(v.p, -__builtin_huge_val()); return r; }

__global__ void __launch_bounds__(64) vlr_pathhmm_kernel(PathArgs a) {
    const int64_t pair = blockIdx.x;
    if (pair >= a.n_pairs) return;
    const int lane = threadIdx.x;
    const uint32_t x0 = a.x_offset[pair], y0 = a.y_offset[pair];
    const int len_x = (int)(a.x_offset[pair + 1] - x0), len_y = (int)(a.y_offset[pair + 1] - y0);
    const double NINF = -__builtin_huge_val();
    if (len_y > 128 || len_y <= 0 || len_x <= 0) {
        if (lane == 0) a.ln_prob[pair] = (len_x <= 0 || len_y <= 0) ? NINF : __builtin_nan("");
        return;
    }
    int yb[2];
    double l_match[2], l_mis[2], l_ins[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * lane + r;
        const bool rowon = j < len_y;
        const int q = rowon ? a.y_quals[y0 + j] : 0;
        yb[r] = rowon ? up(a.y_bases[y0 + j]) : 0;
        const double lm = -(double)q * 2.302585092994046 / 10.0;  // ln P(miscall)
        l_ins[r] = lm;
        l_mis[r] = lm + log(0.3333);                               // PROB_CONFUSION, pairhmm.rs:22-24
        l_match[r] = (lm < -0.693) ? log1p(-exp(lm)) : log(-expm1(lm));
    }
    // column -1: read bases inserted before the first allele base — insertion chain from the start (prev None: gap_x, mod.rs:640-647)
    // pre[j] = gap_x + ins_0 + sum_{k=1..j} (reopen_x + ins_k): inclusive prefix sums over the rows (two per lane)
    double pre[2];
    {
        const double t0 = (lane == 0 ? a.gap_x : a.reopen_x) + l_ins[0], t1 = a.reopen_x + l_ins[1];
        double s = t0 + t1;  // lane sum
        double inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        const double excl = inc - s;
        pre[0] = excl + t0; pre[1] = excl + t0 + t1;
    }
    DP M1[2], D1[2], I1[2], Mt[2], Dt[2], It[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int j = 2 * lane + r;
        M1[r] = {kBig, NINF}; D1[r] = {kBig, NINF};
        I1[r] = {(unsigned)(j + 1), pre[r]};                      // (j, -1)
        Mt[r] = {kBig, NINF}; Dt[r] = {kBig, NINF};
    }
    It[1] = {(unsigned)(2 * lane + 1), pre[0]};                   // (j-1, -1) of the lane's second row = its first row
    {
        const double up1 = __shfl_up(pre[1], 1);
        It[0] = {lane == 0 ? kBig : (unsigned)(2 * lane), lane == 0 ? NINF : up1};  // row 0: the start row, handled below
    }
    const int last_row = len_y - 1;
    const int lr = last_row & 1;
    const bool owner_lane = lane == (last_row >> 1);
    DP best = {kBig, NINF};
    const int nsteps = len_x + len_y - 1;
    int xchunk = 0, xb0 = 0, xb1 = 0;
    for (int d = 0; d < nsteps; ++d) {
        if ((d & 63) == 0) {
            const int i = d + lane;
            xchunk = (i < len_x) ? up(a.x_bases[x0 + i]) : 0;
        }
        const int xnew = __builtin_amdgcn_readlane(xchunk, d & 63);
        const int prev1 = xb1;
        xb1 = xb0;
        xb0 = (int)shr1((unsigned)prev1, (unsigned)xnew);
        DP Mu[2], Du[2], Iu[2];
        Mu[0] = dp_shr1(M1[1]); Du[0] = dp_shr1(D1[1]); Iu[0] = dp_shr1(I1[1]);
        Mu[1] = M1[0]; Du[1] = D1[0]; Iu[1] = I1[0];
        DP Mn[2], Dn[2], In[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 2 * lane + r;
            const int i = d - j;
            const bool incol = (unsigned)i < (unsigned)len_x;
            const int xb = r == 0 ? xb0 : xb1;
            const bool is_match = xb == yb[r];
            const unsigned mm = is_match ? 0u : 1u;
            const double emit = is_match ? l_match[r] : l_mis[r];
            // match / substitution from (j-1, i-1)
            DP m = dp_best(dp_best(dp_step(Mt[r], mm, a.no_gap + emit), dp_step(Dt[r], mm, a.close_y + emit)), dp_step(It[r], mm, a.close_x + emit));
            // deletion (allele base alone, prob_emit_x = 1) from (j, i-1)
            DP dl = dp_best(dp_best(dp_step(M1[r], 1u, a.gap_y), dp_step(D1[r], 1u, a.reopen_y)), dp_step(I1[r], 1u, a.close_x + a.gap_y));
            // insertion (read base alone) from (j-1, i)
            DP in = dp_best(dp_best(dp_step(Mu[r], 1u, a.gap_x + l_ins[r]), dp_step(Iu[r], 1u, a.reopen_x + l_ins[r])), dp_step(Du[r], 1u, a.close_y + a.gap_x + l_ins[r]));
            if (j == 0) {  // the start row above row 0: first operation, no transition term (prev None, mod.rs:598-647)
                const DP sm = {mm, emit}, si = {1u, a.gap_x + l_ins[r]};
                m = dp_best(m, sm);
                in = dp_best(in, si);
            }
            const DP dead = {kBig, NINF};
            Mn[r] = incol ? m : dead; Dn[r] = incol ? dl : dead;
            In[r] = incol ? in : I1[r];  // before its first column a row keeps the insertion chain of column -1
            if (!incol && i >= len_x) In[r] = dead;
        }
        {
            const int i = d - last_row;
            const bool hit = owner_lane && (unsigned)i < (unsigned)len_x;
            const DP e = dp_best(Mn[lr], In[lr]);  // an optimal semiglobal alignment does not end in a deletion
            if (hit) best = dp_best(best, e);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) { Mt[r] = Mu[r]; Dt[r] = Du[r]; It[r] = Iu[r]; M1[r] = Mn[r]; D1[r] = Dn[r]; I1[r] = In[r]; }
    }
    if (owner_lane) a.ln_prob[pair] = best.d >= kBig ? NINF : best.p;
}

}  // namespace vlr

extern "C" int vlr_launch_pathhmm_kernel(const vlr_realign_batch_desc* b, double* ln_prob, void* stream) {
    using namespace vlr;
    if (b->n_pairs <= 0) return 0;
    PathArgs a;
    a.n_pairs = b->n_pairs; a.x_offset = b->x_offset; a.x_bases = b->x_bases; a.y_offset = b->y_offset; a.y_bases = b->y_bases;
    a.y_quals = b->y_quals; a.ln_prob = ln_prob;
    // PathHMMRealigner::new (realignment/mod.rs:560-584)
    const double gx = exp(b->gap[0]), gy = exp(b->gap[1]), gxe = exp(b->gap[2]), gye = exp(b->gap[3]);
    a.gap_x = b->gap[0]; a.gap_y = b->gap[1];
    a.no_gap = log(1.0 - (gx + gy));
    a.close_x = log(1.0 - gxe); a.close_y = log(1.0 - gye);
    a.reopen_x = log(gxe + (1.0 - gxe) * gx); a.reopen_y = log(gye + (1.0 - gye) * gy);
    hipLaunchKernelGGL(vlr_pathhmm_kernel, dim3((unsigned)b->n_pairs), dim3(64), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// hop[16] = ln {hop_x[A,C,G,T] (prob_seq_homopolymer), hop_y[..] (prob_ref_homopolymer), hop_x_extend[..], hop_y_extend[..]}
extern "C" int vlr_launch_homopoly_kernel(const vlr_realign_batch_desc* b, const double* hop, double* ln_prob, void* stream) {
    using namespace vlr;
    if (b->n_pairs <= 0) return 0;
    HomopolyArgs h;
    RealignArgs& a = h.r;
    a.n_pairs = b->n_pairs; a.x_offset = b->x_offset; a.x_bases = b->x_bases; a.y_offset = b->y_offset; a.y_bases = b->y_bases;
    a.y_quals = b->y_quals; a.max_edit_dist = b->max_edit_dist; a.ln_prob = ln_prob;
    const double gx = exp(b->gap[0]), gy = exp(b->gap[1]), gxe = exp(b->gap[2]), gye = exp(b->gap[3]);
    a.pgx = gx; a.pgy = gy; a.pgxe = gxe; a.pgye = gye;
    a.pn = 1.0 - (gx + gy); a.pnx = 1.0 - gxe; a.pny = 1.0 - gye;
    for (int k = 0; k < 4; ++k) { h.hx[k] = exp(hop[k]); h.hy[k] = exp(hop[4 + k]); h.hxe[k] = exp(hop[8 + k]); h.hye[k] = exp(hop[12 + k]); }
    hipLaunchKernelGGL(vlr_homopoly_kernel, dim3((unsigned)b->n_pairs), dim3(64), 0, (hipStream_t)stream, h);
    return (int)hipGetLastError();
}

extern "C" int vlr_launch_realign_kernel(const vlr_realign_batch_desc* b, double* ln_prob, void* stream) {
    using namespace vlr;
    if (b->n_pairs <= 0) return 0;
    RealignArgs a;
    a.n_pairs = b->n_pairs; a.x_offset = b->x_offset; a.x_bases = b->x_bases; a.y_offset = b->y_offset; a.y_bases = b->y_bases;
    a.y_quals = b->y_quals; a.max_edit_dist = b->max_edit_dist; a.ln_prob = ln_prob;
    // GapParamCache of the pair HMM: P(no gap) = 1 - (P(gap x) + P(gap y)); leaving a gap state: 1 - P(extend)
    const double gx = exp(b->gap[0]), gy = exp(b->gap[1]), gxe = exp(b->gap[2]), gye = exp(b->gap[3]);
    a.pgx = gx; a.pgy = gy; a.pgxe = gxe; a.pgye = gye;
    a.pn = 1.0 - (gx + gy); a.pnx = 1.0 - gxe; a.pny = 1.0 - gye;
    static const bool single = getenv("VLR_REALIGN_SINGLE") != nullptr;  // tuning / comparison knob
    if (single) hipLaunchKernelGGL(vlr_realign_kernel, dim3((unsigned)b->n_pairs), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(vlr_realign_kernel2, dim3((unsigned)((b->n_pairs + 1) / 2)), dim3(64), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int vlr_launch_edit_kernel(const vlr_realign_batch_desc* b, int32_t* dist, int32_t* end, int32_t* n_hits, void* stream) {
    using namespace vlr;
    if (b->n_pairs <= 0) return 0;
    EditArgs a;
    a.n_pairs = b->n_pairs; a.x_offset = b->x_offset; a.x_bases = b->x_bases; a.y_offset = b->y_offset; a.y_bases = b->y_bases;
    a.dist = dist; a.end = end; a.n_hits = n_hits;
    hipLaunchKernelGGL(vlr_edit_kernel, dim3((unsigned)b->n_pairs), dim3(64), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
