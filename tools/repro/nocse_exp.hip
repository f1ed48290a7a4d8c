// Minimal reproducer for the "-mllvm -disable-machine-cse builds a wrong kernel" observation of round 1
// (profiles/r01_d_final.md).  The engine kernel built with that flag flags EVERY locus with VLR_LOCUS_UNDERFLOW:
// exp() of a finite log-probability comes back as 0.  This file isolates the f64 exp / expm1 calls of the
// coefficient pass (vlr_kernels.hip, phase B) in a 20-line kernel.
// Mechanism (AMD clang 22 / ROCm 7.2, gfx950): without MachineCSE a 64-bit floating-point constant that is used as a scalar
// operand is materialised as `s_mov_b64 s[0:1], 0x7ff0000000000000`.  gfx950 has no 64-bit literals: the object encoder
// keeps the low 32 bits (`s_mov_b64 s[0:1], 0`, BE8001FF 00000000) and the assembler rejects the textual form
// ("invalid operand for instruction").  Here `x != +inf` turns into `x != 0`; in the engine the range thresholds of exp()
// are hit, so exp(x) = 0 for every negative x.  The default build splits such constants into two s_mov_b32.
// tests/test_build_hygiene.py greps the shipped build's ISA for the pattern.
//   hipcc --offload-arch=gfx950 -O3 nocse_exp.hip -o nocse_exp_ok
//   hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-cse nocse_exp.hip -o nocse_exp_nocse
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ in, double* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = in[i];
    out[3 * i + 0] = exp(x);
    out[3 * i + 1] = -expm1(x);
    out[3 * i + 2] = log1p(exp(x));
}

int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = -0.001f - 30.0f * (float)i / n;
    h[7] = 800.0f;  // exp -> +inf, log1p(+inf) must stay +inf: the miscompiled build compares against 0 instead of +inf
    h[9] = -800.0f;
    float* din; double* dout;
    hipMalloc(&din, n * sizeof(float)); hipMalloc(&dout, 3 * n * sizeof(double));
    hipMemcpy(din, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, din, dout, n);
    std::vector<double> o(3 * n);
    hipMemcpy(o.data(), dout, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    int bad = 0;
    double worst = 0;
    for (int i = 0; i < n; ++i) {
        double x = h[i];
        double r[3] = {std::exp(x), -std::expm1(x), std::log1p(std::exp(x))};
        for (int j = 0; j < 3; ++j) {
            double e = (o[3 * i + j] == r[j]) ? 0.0 : std::fabs(o[3 * i + j] - r[j]) / std::fabs(r[j]);
            if (!(e < 1e-14)) { if (bad < 5) printf("x=%g fn %d gpu %.17g host %.17g\n", x, j, o[3 * i + j], r[j]); bad++; }
            if (e > worst) worst = e;
        }
    }
    printf("nocse_exp: %d of %d values off, worst relative error %.3g\n", bad, 3 * n, worst);
    return bad ? 1 : 0;
}
