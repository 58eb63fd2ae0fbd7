// Timing ablations of conv_mfma_kernel (results are NOT numerically meaningful with ABL != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I topaz_amd/csrc tools/conv_ablate.hip -o /tmp/conv_ablate && /tmp/conv_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_mfma.h"
using namespace tpz;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class C, int ABL>
float run(const ConvArgs& a, dim3 grid, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<C, 0, ABL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((conv_mfma_kernel<C, 0, ABL>), grid, dim3(256), C::LDS_BYTES, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_mfma_kernel<C, 0, ABL>), grid, dim3(256), C::LDS_BYTES, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(e));
    return ms / iters;
}

template <class C>
int bench(const char* name, int cin, int cout, int H) {
    const int span = C::D * (C::K - 1);
    const int Ho = H - span;
    float *in, *w, *out, *zeros, *bias;
    const size_t n_in = (size_t)cin * H * H, n_out = (size_t)cout * Ho * Ho;
    const int n_cog = (cout + C::MT - 1) / C::MT, n_chunks = (cin + C::NCH - 1) / C::NCH;
    const size_t n_w = (size_t)n_cog * n_chunks * C::W_CHUNK;
    CHK(hipMalloc(&in, n_in * 4)); CHK(hipMalloc(&w, n_w * 4)); CHK(hipMalloc(&out, n_out * 4));
    CHK(hipMalloc(&zeros, 256)); CHK(hipMalloc(&bias, cout * 4));
    CHK(hipMemset(zeros, 0, 256)); CHK(hipMemset(bias, 0, cout * 4));
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t o = 0; o < n_in; o += h.size()) CHK(hipMemcpy(in + o, h.data(), std::min(h.size(), n_in - o) * 4, hipMemcpyHostToDevice));
    for (size_t o = 0; o < n_w; o += h.size()) CHK(hipMemcpy(w + o, h.data(), std::min(h.size(), n_w - o) * 4, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.in = in; a.wpk = w; a.bias = bias; a.out = out; a.zeros = zeros;
    a.Cin = a.Cin1 = cin; a.Din = a.D1 = 1; a.Hin = a.H1 = H; a.Win = a.W1 = H;
    a.cs1 = (long long)H * H; a.ps1 = a.cs1; a.pitch1 = H;
    a.cog_inner = 1; a.os = 1; a.Dfull = 1; a.Hfull = Ho; a.Wfull = Ho; a.Cout = cout; a.Dout = 1; a.Hout = Ho; a.Wout = Ho; a.n_chunks = n_chunks; a.slope = 0.f;
    a.tiles_x = (Ho + C::TW - 1) / C::TW;
    a.tiles_y = (Ho + C::TH * C::D - 1) / (C::TH * C::D) * C::D;
    a.tiles_z = 1;
    dim3 grid(a.tiles_x, a.tiles_y, n_cog);
    const double tf = 2.0 * cout * cin * C::K * C::K * (double)Ho * Ho / 1e12;
    printf("%s cin=%d cout=%d out=%d^2 (%.2f TFLOP) LDS=%d B\n", name, cin, cout, Ho, tf, C::LDS_BYTES);
#define RUN(ABL, label) { float ms = run<C, ABL>(a, grid, 2); printf("  %-44s %8.3f ms  %6.1f TF/s\n", label, ms, tf / (ms * 1e-3)); }
    RUN(0, "baseline");
    RUN(0, "baseline (again)");
    RUN(16, "static priority on odd wave slots");
    { ConvArgs b = a; b.xcd_swizzle = 1; float ms = run<C, 0>(b, grid, 2);
      printf("  %-44s %8.3f ms  %6.1f TF/s\n", "XCD-aware tile order", ms, tf / (ms * 1e-3)); }
    {
        ConvArgs b = a;
        b.stagger_first = 512;
        // half a workgroup lifetime: n_stages*STEPS*MW*NW MFMAs * 32 cycles (x2 sharing, /2 half) in 8128-cycle sleeps
        b.stagger_sleeps = (int)((long long)n_chunks * C::SPG * C::STEPS * C::MW * C::NW * 32 / 8128);
        float ms = run<C, 0>(b, grid, 2);
        printf("  %-44s %8.3f ms  %6.1f TF/s\n", "phase stagger (odd slot sleeps T/2 once)", ms, tf / (ms * 1e-3));
        b.stagger_sleeps /= 2;
        ms = run<C, 0>(b, grid, 2);
        printf("  %-44s %8.3f ms  %6.1f TF/s\n", "phase stagger (T/4)", ms, tf / (ms * 1e-3));
    }
    RUN(2, "no per-stage DMA issue");
    RUN(4, "no per-stage barrier");
    RUN(8, "no LDS fragment reads after step 0");
    RUN(2 | 4, "no DMA, no barrier");
    RUN(2 | 4 | 8, "no DMA, no barrier, no LDS reads (MFMA only)");
    hipFree(in); hipFree(w); hipFree(out); hipFree(zeros); hipFree(bias);
    return 0;
}

int main() {
    if (bench<ConvCfg<5, 4, 128, 1, 8, 32, 1, 1, false, 2>>("K5 D4 MT128 RPS1", 128, 256, 2064)) return 1;
    if (bench<ConvCfg<3, 8, 128, 1, 8, 32, 1, 3, false, 2>>("K3 D8 MT128", 128, 128, 2064)) return 1;
    if (bench<ConvCfg<3, 2, 64, 1, 16, 32, 1, 3, false, 2>>("K3 D2 MT64", 64, 64, 2052)) return 1;
    if (bench<ConvCfg<5, 1, 64, 1, 16, 32, 1, 1, false, 2>>("K5 D1 MT64 (x4 DMA)", 96, 64, 2028)) return 1;
    if (bench<ConvCfg<5, 1, 64, 1, 16, 32, 1, 1, false, 2>>("K5 D1 MT64 (x1 DMA: width % 4 != 0)", 96, 64, 2030)) return 1;
    if (bench<ConvCfg<3, 1, 96, 1, 8, 32, 1, 3, false, 2>>("K3 D1 MT96 (x4)", 144, 96, 1016)) return 1;
    if (bench<ConvCfg<3, 1, 96, 1, 8, 32, 1, 3, false, 2>>("K3 D1 MT96 (x1)", 144, 96, 1018)) return 1;
    return 0;
}
