// vlr_fdr.hip — gfx950 kernels of the Bayesian FDR threshold search:
//   /root/reference/src/filtration/fdr.rs:107-141 (posterior distribution in descending order, posterior error probabilities,
//   bio::stats::bayesian::expected_fdr = running mean of the PEPs, largest PEP whose expected FDR is <= alpha without
//   letting equal PEPs cross the boundary).  The collection of the per-variant probabilities (utils/mod.rs:177-270) and the
//   second, filtering pass (utils/mod.rs:288-374) are record I/O and stay on the host (varlociraptor_amd/fdr.py).
//
// Three steps, all HBM-bound integer/f64 streaming work (no MFMA):
//   1. sort the n ln-probabilities in descending order: bitonic network over the array padded to a power of two — the
//      sub-sequences that fit a workgroup (2048 keys = 16 kB of LDS) are merged inside LDS, only strides >= 2048 go through
//      HBM, one coalesced read-modify-write of the array per step;
//   2. PEP_i = 1 - e^{p_i} (after the optional `smart` conversion p -> ln(1 - e^p)) and its inclusive prefix sum: one
//      workgroup-level scan (wave64 DPP-free shuffles + LDS), block offsets scanned by a single workgroup, then applied;
//   3. expected FDR_i = sum_{k<=i} PEP_k / (i + 1) compared with alpha; the last admissible index is an atomic max.
// The reference accumulates in log space (ln_cumsum_exp); the linear f64 prefix sum differs in the last bits only.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlr {

constexpr int kSortBlock = 2048;  // keys per workgroup in the LDS phases (1024 threads, two keys each)

__device__ __forceinline__ void cmpx_desc(double& a, double& b, bool desc) {
    // NaN never occurs (the host filters); -inf pads sort to the end of a descending run
    const bool sw = desc ? (a < b) : (a > b);
    const double t = a;
    a = sw ? b : a;
    b = sw ? t : b;
}

// all steps with stride j < kSortBlock of the merges of size k_lo .. k_hi (k_hi <= kSortBlock: the complete local sort;
// otherwise only the tail of one global merge of size k): data[block * 2048 ...] in LDS
__global__ void __launch_bounds__(1024) fdr_bitonic_local(double* data, unsigned long long k_begin, unsigned long long k_end) {
    __shared__ double s[kSortBlock];
    const unsigned long long base = (unsigned long long)blockIdx.x * kSortBlock;
    const int t = threadIdx.x;
    s[t] = data[base + t];
    s[t + 1024] = data[base + t + 1024];
    __syncthreads();
    for (unsigned long long k = k_begin; k <= k_end; k <<= 1) {
        unsigned long long j0 = (k >> 1) < (unsigned long long)(kSortBlock >> 1) ? (k >> 1) : (unsigned long long)(kSortBlock >> 1);
        for (unsigned long long j = j0; j > 0; j >>= 1) {
            // thread t handles the pair (i, i + j) with i = 2 j (t / j) + (t % j)
            const unsigned i = (unsigned)(2 * j * (t / j) + (t % j));
            const unsigned long long gi = base + i;
            const bool desc = ((gi & k) == 0);  // descending overall: blocks with bit k clear sort descending
            double a = s[i], b = s[i + j];
            cmpx_desc(a, b, desc);
            s[i] = a; s[i + j] = b;
            __syncthreads();
        }
    }
    data[base + t] = s[t];
    data[base + t + 1024] = s[t + 1024];
}

// one step (k, j) with j >= kSortBlock through HBM
__global__ void __launch_bounds__(256) fdr_bitonic_global(double* data, unsigned long long n_half, unsigned long long k, unsigned long long j) {
    const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_half) return;
    const unsigned long long i = 2 * j * (t / j) + (t % j);
    const bool desc = ((i & k) == 0);
    double a = data[i], b = data[i + j];
    cmpx_desc(a, b, desc);
    data[i] = a; data[i + j] = b;
}

__device__ __forceinline__ double ln_one_minus_exp(double p) {  // bio LogProb::ln_one_minus_exp
    if (p < -0.693) return log1p(-exp(p));
    return log(-expm1(p));
}

// PEPs (log and linear) + per-block inclusive scan of the linear PEPs; block totals to `block_sum`
__global__ void __launch_bounds__(1024) fdr_pep_scan(const double* sorted, long long n, int smart, double* prob_out, double* pep_ln, double* cum,
                                                     double* block_sum) {
    __shared__ double wsum[16];
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    double p = (i < n) ? sorted[i] : 0.0;
    if (smart && i < n) p = ln_one_minus_exp(p);  // fdr.rs:110-114
    const double pl = (i < n) ? ln_one_minus_exp(p) : -__builtin_huge_val();
    double v = (i < n) ? -expm1(p) : 0.0;         // PEP = 1 - e^p, linear
    if (i < n) { prob_out[i] = p; pep_ln[i] = pl; }
    // inclusive scan inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    double off = 0.0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    v += off;
    if (i < n) cum[i] = v;
    if (threadIdx.x == 1023) block_sum[blockIdx.x] = v;
}

// exclusive scan of the block totals (sequential over chunks of 1024: n / 1024 totals, a few thousand at most)
__global__ void __launch_bounds__(1024) fdr_scan_blocks(double* block_sum, long long nb) {
    __shared__ double wsum[16];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (long long b0 = 0; b0 < nb; b0 += 1024) {
        const long long i = b0 + threadIdx.x;
        const double x = (i < nb) ? block_sum[i] : 0.0;
        double v = x;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        double off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < nb) block_sum[i] = off + v - x;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + v;
        __syncthreads();
    }
}

// expected FDR and the boundary search (fdr.rs:124-141): out[0] = index of the last admissible entry (or -1)
__global__ void __launch_bounds__(1024) fdr_search(const double* pep_ln, const double* cum, const double* block_off, long long n, double alpha_ln,
                                                   long long* best, double* fdr0) {
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    if (i >= n) return;
    const double c = cum[i] + block_off[blockIdx.x];
    double f = log(c) - log((double)(i + 1));  // bio expected_fdr: ln_cumsum_exp - ln(rank), capped at ln 1
    f = f > 0.0 ? 0.0 : f;
    if (i == 0) *fdr0 = f;
    const bool ok = f <= alpha_ln && (i == 0 || pep_ln[i] != pep_ln[i - 1]);
    if (ok) atomicMax((unsigned long long*)best, (unsigned long long)(i + 1));  // 0 = none
    // The prefix sums are linear here and log-space (sequential ln_add_exp) in the reference: an expected FDR within rounding
    // of alpha could land on the other side of the comparison.  Such entries are reported; the host then repeats the search
    // with the reference's own accumulation (vlr_fdr_threshold), so the decision never depends on the summation order.
    if (fabs(f - alpha_ln) <= 1e-9 * fmax(1.0, fabs(alpha_ln))) best[2] = 1;
}

}  // namespace vlr

// host-callable driver; returns a hipError_t as int.  `work` holds npad + 3 n + nblocks + 4 doubles.
extern "C" int vlr_launch_fdr(double* keys, long long n, long long npad, int smart, double alpha_ln, double* work, long long* best, double* fdr0,
                              void* stream_) {
    using namespace vlr;
    hipStream_t st = (hipStream_t)stream_;
    const unsigned long long N = (unsigned long long)npad;
    // 1. sort (descending)
    const unsigned nblk = (unsigned)(N / kSortBlock);
    hipLaunchKernelGGL(fdr_bitonic_local, dim3(nblk), dim3(1024), 0, st, keys, 2ull, (unsigned long long)kSortBlock);
    for (unsigned long long k = 2ull * kSortBlock; k <= N; k <<= 1) {
        for (unsigned long long j = k >> 1; j >= (unsigned long long)kSortBlock; j >>= 1)
            hipLaunchKernelGGL(fdr_bitonic_global, dim3((unsigned)((N / 2 + 255) / 256)), dim3(256), 0, st, keys, N / 2, k, j);
        hipLaunchKernelGGL(fdr_bitonic_local, dim3(nblk), dim3(1024), 0, st, keys, k, k);  // strides below 2048 of this merge
    }
    // 2. PEPs + prefix sums
    double* prob = work;
    double* pep = prob + n;
    double* cum = pep + n;
    double* bsum = cum + n;
    const long long nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(fdr_pep_scan, dim3((unsigned)nb), dim3(1024), 0, st, keys, n, smart, prob, pep, cum, bsum);
    hipLaunchKernelGGL(fdr_scan_blocks, dim3(1), dim3(1024), 0, st, bsum, nb);
    // 3. search
    hipLaunchKernelGGL(fdr_search, dim3((unsigned)nb), dim3(1024), 0, st, pep, cum, bsum, n, alpha_ln, best, fdr0);
    return (int)hipGetLastError();
}
