// Sustained f16 MFMA rate and shader clock by instruction shape (the chip runs at its power limit under MFMA load):
//   v_mfma_f32_16x16x32_f16  (what conv_split.h issues)  vs  v_mfma_f32_32x32x16_f16 (half the operand reads per flop).
// Register operands only -- no LDS, no memory: the ceiling a K loop could reach if everything else were free.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_power.hip -o tools/_bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void mfma_loop(const f16x8* src, float* out, int iters, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(threadIdx.x + 64 * i) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x * 3 + 17 * i) & 4095];
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8][4];
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][3];
    } else {
        f32x16 acc[4][2];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // same flops per iteration: 4 x 2 tiles of 32x32, K = 2 x 16
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * m + k], b[2 * n + k], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][15];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        atomicAdd(&clk[0], __builtin_readcyclecounter() - c0);
        atomicAdd(&clk[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int SHAPE>
static void run(const char* label, const f16x8* src, float* out, unsigned long long* clk, int iters, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(clk, 0, 16);
        hipEventRecord(e0);
        mfma_loop<SHAPE><<<wgs, 512>>>(src, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        const double flop = (double)wgs * 8 * iters * 32 * (2.0 * 16 * 16 * 32);
        printf("  %-28s %8.3f ms  %7.1f TFLOP/s f16 (%.1f fp32-equivalent at 3 MFMAs/MAC)   clock %.3f GHz\n", label, ms,
               flop / (ms * 1e-3) * 1e-12, flop / (ms * 1e-3) * 1e-12 / 3, c[1] ? 0.1 * (double)c[0] / (double)c[1] : 0.0);
    }
}

int main(int argc, char** argv) {
    const int iters = 20000, wgs = 256 * 4;
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    for (auto& v : h) v = zeros ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    f16x8* src; float* out; unsigned long long* clk;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, (size_t)wgs * 512 * 4); hipMalloc(&clk, 16);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("%s operands, %d workgroups x 8 waves, %d x 32 MFMAs per wave\n", zeros ? "zero" : "random", wgs, iters);
    run<16>("v_mfma_f32_16x16x32_f16", src, out, clk, iters, wgs);
    run<32>("v_mfma_f32_32x32x16_f16", src, out, clk, iters, wgs);
    run<16>("v_mfma_f32_16x16x32_f16", src, out, clk, iters, wgs);
    return 0;
}
