// Reproducers of the two asm-statement defects behind the build-matrix deviations of rounds 4-5 (fixed in round 6; DESIGN.md "Build
// matrix", csrc/vlr_kernels.hip "park_sd" / "fresh_sd").  Neither is a compiler bug: both are things a compiler cannot check in a template.
//   hipcc --offload-arch=gfx950 -O3 tools/repro/asm_lane_read_hazard.hip -o /tmp/asm_repro && /tmp/asm_repro
// (1) lane read inside an asm statement: gfx950 needs one wait state between a VALU instruction that writes a VGPR and a
//     v_readfirstlane / v_readlane of that VGPR.  The hazard recogniser inserts it in front of its OWN lane reads; a template is opaque.
//     Here the f64 add is the instruction in front of the statement, and the low word comes back stale.
// (2) two-instruction template without early-clobber outputs: the first instruction writes %0 before the second reads %3, so %0 must
//     not share a register with %3 — "=s" allows exactly that when the input dies at the statement ("=&s" forbids it).  Whether the
//     allocator does it depends on everything around; the kernel below only shows the constraint the old fresh_sd relied on by luck.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__global__ void lane_read_in_asm(const double* a, double* bad, double* good, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[blockIdx.x];            // wave-uniform input
    const double v = 1.0 - x;                   // v_add_f64: the producer
    int lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(v)), "v"(__double2hiint(v)));
    bad[i] = __hiloint2double(hi, lo);
    good[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__global__ void moves_without_early_clobber(const int* in, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int vlo = __builtin_amdgcn_readfirstlane(in[2 * blockIdx.x]), vhi = __builtin_amdgcn_readfirstlane(in[2 * blockIdx.x + 1]);
    int lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "s"(vlo), "s"(vhi));   // legal for the compiler: %0 == %3
    out[i] = __hiloint2double(hi, lo);
}

int main() {
    const int blocks = 4096, n = blocks * 64;
    double *a, *bad, *good, *out; int* in;
    (void)hipMallocManaged(&a, blocks * 8); (void)hipMallocManaged(&bad, n * 8); (void)hipMallocManaged(&good, n * 8); (void)hipMallocManaged(&out, n * 8); (void)hipMallocManaged(&in, blocks * 8);
    for (int b = 0; b < blocks; ++b) { a[b] = 0.3 + 1e-9 * b; const double r = 1.0 - a[b]; uint64_t u; memcpy(&u, &r, 8); in[2 * b] = (int)(uint32_t)u; in[2 * b + 1] = (int)(uint32_t)(u >> 32); }
    hipLaunchKernelGGL(lane_read_in_asm, dim3(blocks), dim3(64), 0, 0, a, bad, good, n);
    hipLaunchKernelGGL(moves_without_early_clobber, dim3(blocks), dim3(64), 0, 0, in, out, n);
    (void)hipDeviceSynchronize();
    int n_bad = 0, n_good = 0, n_mov = 0;
    for (int i = 0; i < n; ++i) { const double want = 1.0 - a[i / 64]; n_bad += bad[i] != want; n_good += good[i] != want; n_mov += out[i] != want; }
    printf("lane read in asm behind its producer: %d of %d values wrong (builtin lane read: %d wrong)\n", n_bad, n, n_good);
    printf("two moves without early-clobber outputs: %d of %d values wrong (0 unless the allocator shares %%0 with %%3 here)\n", n_mov, n);
    return 0;
}
