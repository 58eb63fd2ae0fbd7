// Diagnostics: the matrix-pipe rate the board SUSTAINS.  A register-resident loop of the instruction the 2xf16 kernels issue
// (v_mfma_f32_16x16x32_f16), two waves per SIMD on every CU, no memory traffic: with operands of full entropy the power
// management holds the shader clock near 1.9 GHz (not the 2.4 GHz the dense peak is quoted at); with all-zero operands the same
// loop runs at 2.4 GHz.  tpz_prof_mfma_sustained reports the rate, bench.py quotes the dominant kernel against it next to the
// nominal roofline (profiles/r04_mfma_sustained.txt).
#include "kernels_misc.h"

namespace tpz {

typedef _Float16 spin_half8 __attribute__((ext_vector_type(8)));
typedef float spin_float4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_spin_kernel(const spin_half8* __restrict__ src, float* __restrict__ out, int iters,
                                                        unsigned long long* __restrict__ ticks) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    spin_half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(t * 8 + i) & 4095];
        b[i] = src[(t * 8 + 4 + i) & 4095];
    }
    spin_float4 c[8] = {};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)     // eight independent accumulators: back-to-back issue, no dependent stall
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(a[u & 3]), "v"(b[(u + 1) & 3]));
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int u = 0; u < 8; ++u) s += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    out[t] = s;
    if (t == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
}

hipError_t launch_mfma_spin(const void* src, float* out, int n_wg, int iters, unsigned long long* ticks, hipStream_t st) {
    hipLaunchKernelGGL(mfma_spin_kernel, dim3(n_wg), dim3(256), 0, st, (const spin_half8*)src, out, iters, ticks);
    return hipGetLastError();
}

}  // namespace tpz
