// Timing ablations of conv_split_kernel (results are NOT numerically meaningful with ABL != 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I topaz_amd/csrc tools/split_ablate.hip -o tools/_run/split_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include "conv_split.h"
using namespace tpz;
#define TPZ_C ,

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// the K-loop schedule the kernel reads (conv_split.h split_make_plan), as the library builds it
template <class C>
const SplitStep* make_plan(const SplitArgs& a) {
    SplitPlanKey k{};
    k.cells_in = a.cells_in; k.cells_in1 = a.cells_in1; k.n_chunks = a.n_chunks; k.has_in2 = a.in2 != nullptr;
    k.vol = 0; k.KZ = 1; k.fold_cells = a.fold_cells; k.fold_tap = a.fold_tap;
    std::vector<SplitStep> h;
    split_make_plan<C>(k, h);
    SplitStep* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(SplitStep)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(SplitStep), hipMemcpyHostToDevice);
    return d;
}

template <class C, int EPI, int ABL, int MODE = 0>
float run(const SplitArgs& a, dim3 grid, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_kernel<C, EPI, ABL, MODE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((conv_split_kernel<C, EPI, ABL, MODE>), grid, dim3(C::THREADS), C::LDS_BYTES, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv_split_kernel<C, EPI, ABL, MODE>), grid, dim3(C::THREADS), C::LDS_BYTES, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(e));
    return ms / iters;
}

static bool g_lat = false, g_exp = false;
static int g_issuer_half = 0;      // quick(): SplitArgs::issuer_half of the timed runs (the library sets 1 on the 8-wave MT >= 96 tiles)
template <class C, int EPI>
int bench(const char* name, int cin, int cout, int H) {
    const int span = C::D * (C::K - 1);
    const int Ho = H - span;
    const size_t cells_in = (cin + 7) / 8, cells_out = (cout + 7) / 8;
    const size_t n_in = cells_in * 8 * H * H, n_out = cells_out * 8 * (size_t)Ho * Ho;
    const int n_cog = (cout + C::MT - 1) / C::MT, n_chunks = (int)((cells_in + C::CC - 1) / C::CC);
    const int n_st = C::CONT ? C::cont_stages((int)cells_in) : n_chunks * C::NSTEP;
    const size_t n_w = (size_t)n_cog * n_st * C::W_STEP_BYTES / 4;
    float *in, *w, *out, *res, *zeros, *vec;
    unsigned* flag;
    CHK(hipMalloc(&in, n_in * 4)); CHK(hipMalloc(&w, n_w * 4)); CHK(hipMalloc(&out, n_out * 4)); CHK(hipMalloc(&res, n_out * 4));
    CHK(hipMalloc(&zeros, 256)); CHK(hipMalloc(&vec, (cout + 512) * 4)); CHK(hipMemset(vec, 0, (cout + 512) * 4)); CHK(hipMalloc(&flag, 256));
    CHK(hipMemset(zeros, 0, 256)); CHK(hipMemset(flag, 0, 256));
    // f16 bit patterns of small normal numbers (0x2xxx..0x3xxx ~ 0.01 .. 1)
    std::vector<uint16_t> h(1 << 21);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x2800 + ((i * 2654435761u) >> 20) % 0x1000) | (uint16_t)(((i * 40503u) >> 7) & 1) << 15;
    for (size_t o = 0; o < n_in * 4; o += h.size() * 2) CHK(hipMemcpy((char*)in + o, h.data(), std::min(h.size() * 2, n_in * 4 - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < n_w * 4; o += h.size() * 2) CHK(hipMemcpy((char*)w + o, h.data(), std::min(h.size() * 2, n_w * 4 - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < n_out * 4; o += h.size() * 2) CHK(hipMemcpy((char*)res + o, h.data(), std::min(h.size() * 2, n_out * 4 - o), hipMemcpyHostToDevice));
    std::vector<float> ones(cout, 1e-3f);
    CHK(hipMemcpy(vec, ones.data(), cout * 4, hipMemcpyHostToDevice));
    SplitArgs a{};
    a.in = (const uint4*)in; a.wpk = (const uint4*)w; a.wscale = vec; a.bias = vec; a.out = (uint4*)out; a.res = (const uint4*)res;
    a.post_scale = vec; a.post_shift = vec; a.head_w = vec; a.head_out = out; a.zeros = zeros; a.flag = flag;
    a.cells_in = a.cells_in1 = (int)cells_in; a.Hin = a.H1 = H; a.Win = a.W1 = H; a.Cout = cout; a.cells_out = (int)cells_out; a.Hout = Ho; a.Wout = Ho;
    a.os = 1; a.Hfull = Ho; a.Wfull = Ho; a.Hres = Ho; a.Wres = Ho; a.n_chunks = n_chunks; a.xcd_swizzle = 1;
    a.KZ = 1; a.Din = a.Dout = a.Dfull = a.Dres = 1;     // a 2-D launch, as launch_split fills it in
    a.wy0 = a.wx0 = 0; a.wy1 = Ho; a.wx1 = Ho;          // the whole lattice
    a.cog_inner = EPI == EPI_HEAD ? n_cog : 1;
    a.tiles_x = (Ho + C::TW - 1) / C::TW;
    a.tiles_y = (Ho + C::TH * C::D - 1) / (C::TH * C::D) * C::D;
    a.plan = make_plan<C>(a);
    dim3 grid(a.tiles_x, a.tiles_y, n_cog / a.cog_inner);
    const double tf = 2.0 * cout * cin * C::K * C::K * (double)Ho * Ho / 1e12;
    printf("%s cin=%d cout=%d out=%d^2 (%.2f TFLOP fp32-equivalent) LDS=%d B, %d steps/tile\n", name, cin, cout, Ho, tf, C::LDS_BYTES,
           n_st * a.cog_inner);
#define RUN(ABL, label) { hipMemset(flag, 0, 256); float ms = run<C, EPI, (ABL) | 2048>(a, grid, 6); unsigned long long c[4]; hipMemcpy(c, flag, 32, hipMemcpyDeviceToHost); \
    printf("  %-52s %8.3f ms  %6.1f TF/s   clock %.3f GHz\n", label, ms, tf / (ms * 1e-3), c[2] ? 0.1 * (double)c[1] / (double)c[2] : 0.0); }
    if (g_exp) {
        // round-3 experiments on the plan-driven K loop: DMA form, where in the step the DMA is issued, order of fragment 0
        for (int ih = 0; ih < (C::ISSUER_HALF ? 2 : 1); ++ih) {
            a.issuer_half = ih;
            printf(" issuer_half = %d\n", ih);
            RUN(0, "baseline (buffer DMA, issued at the top of the step)");
            RUN(65536, "round-2 DMA forms (global_load_lds)");
            RUN(131072, "fragment 0: A(1) request first (round-2 order)");
            RUN(0, "baseline (again)");
        }
        return 0;
    }
    if (g_lat) {
        // short list: what the DMA costs, split into issue and waiting for arrival
        if (C::WAVES == 8 && C::MT >= 96) a.issuer_half = 1;           // as the library launches it
        RUN(0, "baseline");
        RUN(0, "baseline (again)");
        RUN(0, "baseline (third)");
        RUN(16384, "DMA issued, its arrival never waited for");
        RUN(16384 | 4, "DMA never waited for, no barrier");
        RUN(2, "no per-step DMA issue");
        RUN(16, "no MFMA (everything else)");
        RUN(16 | 16384, "no MFMA, DMA never waited for");
        RUN(16 | 2, "no MFMA, no DMA issue");
        RUN(16 | 8, "no MFMA, fragment reads for m = 0 only");
        RUN(16 | 4, "no MFMA, no barrier");
        RUN(16 | 2 | 4, "no MFMA, no DMA, no barrier");
        RUN(16 | 2 | 4 | 8, "no MFMA, no DMA, no barrier, few LDS reads");
        RUN(16 | 2 | 4 | 8 | 1, "  ... and no epilogue loads/stores (loop skeleton + prologue)");
        return 0;
    }
    RUN(0, "baseline");
    RUN(0, "baseline (again)");
    { SplitArgs a0 = a; a.issuer_half = 1; RUN(0, "upper half of the waves issues all DMA"); RUN(0, "upper half issues all DMA (again)"); a = a0; }
    RUN(0, "baseline (third)");
    RUN(1, "no epilogue loads/stores");
    RUN(32, "no residual loads");
    RUN(64, "no stores");
    RUN(2, "no per-step DMA issue");
    RUN(128, "no per-step WEIGHT DMA (input DMA kept)");
    RUN(256, "no per-step INPUT DMA (weight DMA kept)");
    RUN(16384, "DMA issued, its arrival never waited for");
    RUN(16384 | 4, "DMA never waited for, no barrier");
    RUN(4, "no per-step barrier");
    RUN(8, "A fragments read once per step");
    RUN(1 | 2, "no epilogue, no DMA");
    RUN(1 | 2 | 4, "no epilogue, no DMA, no barrier");
    RUN(1 | 2 | 4 | 8, "no epilogue, no DMA, no barrier, few LDS reads");
    RUN(16, "no MFMA (everything else)");
    RUN(1 | 16, "no MFMA, no epilogue");
    hipFree(in); hipFree(w); hipFree(out); hipFree(res); hipFree(zeros); hipFree(vec); hipFree(flag);
    return 0;
}

template <class C, int EPI, int ABLX = 0>
int quick(const char* name, int cin, int cout, int H) {
    const int span = C::D * (C::K - 1);
    const int Ho = H - span;
    const size_t cells_in = (cin + 7) / 8, cells_out = (cout + 7) / 8;
    const size_t n_in = cells_in * 8 * H * H, n_out = cells_out * 8 * (size_t)Ho * Ho;
    const int n_cog = (cout + C::MT - 1) / C::MT, n_chunks = (int)((cells_in + C::CC - 1) / C::CC);
    const int n_st = C::CONT ? C::cont_stages((int)cells_in) : n_chunks * C::NSTEP;
    const size_t n_w = (size_t)n_cog * n_st * C::W_STEP_BYTES / 4;
    float *in, *w, *out, *res, *zeros, *vec;
    unsigned* flag;
    CHK(hipMalloc(&in, n_in * 4)); CHK(hipMalloc(&w, n_w * 4)); CHK(hipMalloc(&out, n_out * 4)); CHK(hipMalloc(&res, n_out * 4));
    CHK(hipMalloc(&zeros, 256)); CHK(hipMalloc(&vec, (cout + 512) * 4)); CHK(hipMemset(vec, 0, (cout + 512) * 4)); CHK(hipMalloc(&flag, 256));
    CHK(hipMemset(zeros, 0, 256)); CHK(hipMemset(flag, 0, 256));
    // f16 bit patterns of small normal numbers (0x2xxx..0x3xxx ~ 0.01 .. 1)
    std::vector<uint16_t> h(1 << 21);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x2800 + ((i * 2654435761u) >> 20) % 0x1000) | (uint16_t)(((i * 40503u) >> 7) & 1) << 15;
    for (size_t o = 0; o < n_in * 4; o += h.size() * 2) CHK(hipMemcpy((char*)in + o, h.data(), std::min(h.size() * 2, n_in * 4 - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < n_w * 4; o += h.size() * 2) CHK(hipMemcpy((char*)w + o, h.data(), std::min(h.size() * 2, n_w * 4 - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < n_out * 4; o += h.size() * 2) CHK(hipMemcpy((char*)res + o, h.data(), std::min(h.size() * 2, n_out * 4 - o), hipMemcpyHostToDevice));
    std::vector<float> ones(cout, 1e-3f);
    CHK(hipMemcpy(vec, ones.data(), cout * 4, hipMemcpyHostToDevice));
    SplitArgs a{};
    a.in = (const uint4*)in; a.wpk = (const uint4*)w; a.wscale = vec; a.bias = vec; a.out = (uint4*)out; a.res = (const uint4*)res;
    a.post_scale = vec; a.post_shift = vec; a.head_w = vec; a.head_out = out; a.zeros = zeros; a.flag = flag;
    a.cells_in = a.cells_in1 = (int)cells_in; a.Hin = a.H1 = H; a.Win = a.W1 = H; a.Cout = cout; a.cells_out = (int)cells_out; a.Hout = Ho; a.Wout = Ho;
    a.os = 1; a.Hfull = Ho; a.Wfull = Ho; a.Hres = Ho; a.Wres = Ho; a.n_chunks = n_chunks; a.xcd_swizzle = 1;
    a.KZ = 1; a.Din = a.Dout = a.Dfull = a.Dres = 1;     // a 2-D launch, as launch_split fills it in
    a.wy0 = a.wx0 = 0; a.wy1 = Ho; a.wx1 = Ho;          // the whole lattice
    a.issuer_half = (C::ISSUER_HALF && g_issuer_half) ? 1 : 0;
    a.cog_inner = EPI == EPI_HEAD ? n_cog : 1;
    a.tiles_x = (Ho + C::TW - 1) / C::TW;
    a.tiles_y = (Ho + C::TH * C::D - 1) / (C::TH * C::D) * C::D;
    a.plan = make_plan<C>(a);
    dim3 grid(a.tiles_x, a.tiles_y, n_cog / a.cog_inner);
    const double tf = 2.0 * cout * cin * C::K * C::K * (double)Ho * Ho / 1e12;
    printf("%s cin=%d cout=%d out=%d^2 (%.2f TFLOP fp32-equivalent) LDS=%d B, %d steps/tile\n", name, cin, cout, Ho, tf, C::LDS_BYTES,
           n_st * a.cog_inner);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(flag, 0, 256);
        float ms = run<C, EPI, 2048 | ABLX>(a, grid, 10);
        unsigned long long c[6];
        hipMemcpy(c, flag, 48, hipMemcpyDeviceToHost);
        // cycles per workgroup (11 launches accumulated): whole kernel, prologue (tables + first tile + wait), K loop, epilogue
        const double nwg = 11.0 * grid.x * grid.y * grid.z;
        printf("  run %d  %8.3f ms  %6.1f TF/s   clock %.3f GHz   cycles/tile: total %.0f = prologue %.0f + K loop %.0f (%.0f / step) + epilogue %.0f\n",
               rep, ms, tf / (ms * 1e-3), c[2] ? 0.1 * (double)c[1] / (double)c[2] : 0.0, c[1] / nwg, c[3] / nwg, c[4] / nwg,
               c[4] / nwg / (n_st * a.cog_inner), (c[1] - c[3] - c[4]) / nwg);
    }
    for (int ih = 0; ih < (C::ISSUER_HALF ? 2 : 1); ++ih) {
        // where a step's cycles go (ABL 4096): per wave, averaged over all steps of all tiles
        SplitArgs t = a;
        t.issuer_half = ih;
        hipMemset(flag, 0, 256);
        float ms = run<C, EPI, 2048 | 4096>(t, grid, 10);
        unsigned long long c[16];
        hipMemcpy(c, flag, 128, hipMemcpyDeviceToHost);
        const double nst = 11.0 * grid.x * grid.y * grid.z * n_st * a.cog_inner;
        printf("  step timeline (issuer_half %d, %.3f ms): wave 0: dma %.0f | frags %.0f | drain+barrier %.0f | last frag %.0f   wave %d: dma %.0f | frags %.0f | drain+barrier %.0f | last frag %.0f\n",
               ih, ms, c[5] / nst, c[6] / nst, c[7] / nst, c[8] / nst, C::WAVES / 2, c[9] / nst, c[10] / nst, c[11] / nst, c[12] / nst);
    }
    if constexpr (EPI != EPI_HEAD) {
        // persistent workgroups (MODE 4): CUs x workgroups-per-CU of them walk the tiles, each prefetching its next tile
        SplitArgs p = a;
        p.n_tiles = (int)(grid.x * grid.y * grid.z);
        const dim3 pgrid(256 * C::WGS_PER_CU, 1, 1);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(flag, 0, 256);
            float ms = run<C, EPI, 2048, 4>(p, pgrid, 10);
            unsigned long long c[6];
            hipMemcpy(c, flag, 48, hipMemcpyDeviceToHost);
            const double nwg = 11.0 * p.n_tiles;
            printf("  persistent run %d  %8.3f ms  %6.1f TF/s   clock %.3f GHz   cycles/tile: total %.0f = prologue %.0f + K loop %.0f (%.0f / step) + epilogue %.0f\n",
                   rep, ms, tf / (ms * 1e-3), c[2] ? 0.1 * (double)c[1] / (double)c[2] : 0.0, c[1] / nwg, c[3] / nwg, c[4] / nwg,
                   c[4] / nwg / n_st, (c[1] - c[3] - c[4]) / nwg);
        }
    }
    hipFree(in); hipFree(w); hipFree(out); hipFree(res); hipFree(zeros); hipFree(vec); hipFree(flag);
    return 0;
}



int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "proj") {
        // 1x1 projection 64 -> 128 at 4136^2 (ResidA.proj): tile shapes, same process
        quick<SplitCfg<1, 1, 128, 16, 16, 4, 8, 1, 1>, EPI_PLAIN>("K1 MT128 8w 16x16 CC4 (current)", 64, 128, 4136);
        quick<SplitCfg<1, 1, 128, 8, 16, 4, 4, 1, 1>, EPI_PLAIN>("K1 MT128 4w 8x16 CC4", 64, 128, 4136);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "d2") {
        // 64-channel dilation-2 layers of ResNet8's first block: 8 waves (one workgroup per CU) vs 4 waves (two per CU)
        quick<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D2 MT64 8w S=2 (current)", 64, 64, 2048);
        quick<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT64 4w S=1", 64, 64, 2048);
        quick<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D2 MT64 8w S=2 RES (current)", 64, 64, 2048);
        quick<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_RES>("K3 D2 MT64 4w S=1 RES", 64, 64, 2048);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "lat") {
        // is the per-step DMA cost its ISSUE or waiting for its ARRIVAL?  (round 2: the issue -- never waiting for the
        // data changes nothing; a dedicated producer wave issuing all of it was 7 - 30 % SLOWER: one wave sustains about
        // one 1-KiB piece per 150 cycles, the pieces have to be spread over many waves)
        g_lat = true;
        if (argc > 2) {        // the narrow tiles only
            bench<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2 (dec1.2 64->32 at 2024^2)", 64, 32, 2028);
            bench<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (U-Net dec 96->96 at 1012^2)", 96, 96, 1014);
            bench<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D2 MT64 8w S=2 (ResNet8 conv0)", 64, 64, 2048);
            return 0;
        }
        bench<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2 (dec1.2 64->32 at 2024^2)", 64, 32, 2028);
        bench<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (U-Net dec 96->96 at 1012^2)", 96, 96, 1014);
        bench<SplitCfg<3, 1, 128, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT128 4w (sub-pixel dec1.0: 104 -> 256 virtual at 1012^2)", 104, 256, 1014);
        bench<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D2 MT64 8w S=2 (ResNet8 conv0)", 64, 64, 2048);
        bench<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D4 MT128 8w S=2 RES cin=64", 64, 128, 2048);
        bench<SplitCfg<3, 8, 128, 16, 32, 2, 8, 3, 1>, EPI_RES>("K3 D8 MT128 8w RES", 128, 128, 2048);
        bench<SplitCfg<5, 4, 128, 16, 32, 2, 8, 5, 1>, EPI_HEAD>("K5 D4 MT128 HEAD", 128, 256, 2048);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "exp") {
        g_exp = true;
        bench<SplitCfg<5, 4, 128, 16, 32, 2, 8, 5, 1>, EPI_HEAD>("K5 D4 MT128 HEAD", 128, 256, 2048);
        bench<SplitCfg<3, 8, 128, 16, 32, 2, 8, 3, 1>, EPI_RES>("K3 D8 MT128 8w RES", 128, 128, 2048);
        bench<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D4 MT128 8w S=2", 128, 128, 2048);
        bench<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT64 4w", 64, 64, 2048);
        bench<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D4 MT64 8w S=2 RES", 64, 64, 2048);
        bench<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (U-Net dec 96->96 at 1012^2)", 96, 96, 1014);
        bench<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2 (dec1.2 64->32 at 2024^2)", 64, 32, 2028);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "c127") {
        // round 6: steps per stage on the 5x5 32 -> 32 tiles of conv127 that have a continuous slot stream (CC = 2: d2, d4)
        quick<SplitCfg<5, 4, 32, 16, 32, 2, 8, 5, 1>, EPI_PLAIN>("K5 D4 MT32 8w 16x32 CC2 S=1 (current)", 32, 32, 2064);
        quick<SplitCfg<5, 4, 32, 16, 32, 2, 8, 5, 2>, EPI_PLAIN>("K5 D4 MT32 8w 16x32 CC2 S=2", 32, 32, 2064);
        quick<SplitCfg<5, 4, 32, 16, 32, 2, 8, 5, 4>, EPI_PLAIN>("K5 D4 MT32 8w 16x32 CC2 S=4", 32, 32, 2064);
        quick<SplitCfg<5, 2, 32, 8, 32, 2, 4, 5, 1>, EPI_PLAIN>("K5 D2 MT32 4w 8x32 CC2 S=1 (current)", 32, 32, 2056);
        quick<SplitCfg<5, 2, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D2 MT32 4w 8x32 CC2 S=2", 32, 32, 2056);
        quick<SplitCfg<5, 2, 32, 16, 32, 2, 8, 5, 1>, EPI_PLAIN>("K5 D2 MT32 8w 16x32 CC2 S=1", 32, 32, 2056);
        quick<SplitCfg<5, 2, 32, 16, 32, 2, 8, 5, 4>, EPI_PLAIN>("K5 D2 MT32 8w 16x32 CC2 S=4", 32, 32, 2056);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "prio") {
        // round 6: static issue priority for one half of the 8-wave workgroups during the K loop
#define PRIO_AB(CFG, E, label, cin, cout, H) { for (int ih = 0; ih < 2; ++ih) { g_issuer_half = ih; printf("issuer_half %d\n", ih); \
            quick<CFG, E>(label " [no priority]", cin, cout, H); quick<CFG, E, 524288>(label " [waves 4-7 prio 1]", cin, cout, H); \
            quick<CFG, E, 1048576>(label " [waves 0-3 prio 1]", cin, cout, H); } g_issuer_half = 0; }
        PRIO_AB(SplitCfg<3 TPZ_C 8 TPZ_C 128 TPZ_C 16 TPZ_C 32 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 1>, EPI_RES, "K3 D8 MT128 8w RES", 128, 128, 2048)
        PRIO_AB(SplitCfg<3 TPZ_C 4 TPZ_C 128 TPZ_C 16 TPZ_C 32 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 2>, EPI_PLAIN, "K3 D4 MT128 8w S=2", 128, 128, 2048)
        PRIO_AB(SplitCfg<5 TPZ_C 4 TPZ_C 128 TPZ_C 16 TPZ_C 32 TPZ_C 2 TPZ_C 8 TPZ_C 5 TPZ_C 1>, EPI_HEAD, "K5 D4 MT128 HEAD", 128, 256, 2048)
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "epi") {
        // round 6: paired 16-byte epilogue stores / residual loads (v_permlane16_swap) against the 8-byte ones (ABL 262144)
#define EPI_AB(CFG, E, label, cin, cout, H) quick<CFG, E, 262144>(label " [8-byte accesses]", cin, cout, H); quick<CFG, E>(label " [16-byte, paired]", cin, cout, H);
        EPI_AB(SplitCfg<3 TPZ_C 8 TPZ_C 128 TPZ_C 16 TPZ_C 32 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 1>, EPI_RES, "K3 D8 MT128 8w RES", 128, 128, 2048)
        EPI_AB(SplitCfg<3 TPZ_C 4 TPZ_C 128 TPZ_C 16 TPZ_C 32 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 2>, EPI_PLAIN, "K3 D4 MT128 8w S=2", 128, 128, 2048)
        EPI_AB(SplitCfg<3 TPZ_C 2 TPZ_C 64 TPZ_C 16 TPZ_C 48 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 1>, EPI_PLAIN, "K3 D2 MT64 8w 16x48", 64, 64, 2052)
        EPI_AB(SplitCfg<3 TPZ_C 4 TPZ_C 64 TPZ_C 16 TPZ_C 48 TPZ_C 2 TPZ_C 8 TPZ_C 3 TPZ_C 1>, EPI_RES, "K3 D4 MT64 8w 16x48 RES", 64, 64, 2056)
        EPI_AB(SplitCfg<3 TPZ_C 1 TPZ_C 96 TPZ_C 8 TPZ_C 32 TPZ_C 2 TPZ_C 4 TPZ_C 3 TPZ_C 1>, EPI_PLAIN, "K3 D1 MT96 4w (U-Net 96->96 at 1012^2)", 96, 96, 1014)
        EPI_AB(SplitCfg<3 TPZ_C 1 TPZ_C 128 TPZ_C 8 TPZ_C 32 TPZ_C 2 TPZ_C 4 TPZ_C 3 TPZ_C 1>, EPI_PLAIN, "K3 D1 MT128 4w (sub-pixel dec1.0)", 104, 256, 1014)
        EPI_AB(SplitCfg<5 TPZ_C 1 TPZ_C 32 TPZ_C 8 TPZ_C 32 TPZ_C 2 TPZ_C 4 TPZ_C 5 TPZ_C 2>, EPI_PLAIN, "K5 D1 MT32 4w S=2 (dec1.2 64->32)", 64, 32, 2028)
        EPI_AB(SplitCfg<3 TPZ_C 1 TPZ_C 48 TPZ_C 8 TPZ_C 32 TPZ_C 2 TPZ_C 4 TPZ_C 3 TPZ_C 1>, EPI_PLAIN, "K3 D1 MT48 4w", 48, 48, 1024)
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "ab") {
        quick<SplitCfg<5, 4, 128, 16, 32, 2, 8, 5, 1>, EPI_HEAD>("K5 D4 MT128 HEAD", 128, 256, 2048);
        quick<SplitCfg<3, 8, 128, 16, 32, 2, 8, 3, 1>, EPI_RES>("K3 D8 MT128 8w RES", 128, 128, 2048);
        quick<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D4 MT128 8w S=2", 128, 128, 2048);
        quick<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 1>, EPI_PLAIN>("K3 D4 MT128 8w S=1 cin=64", 64, 128, 2048);
        quick<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT64 4w", 64, 64, 2048);
        quick<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D4 MT64 8w S=2 RES", 64, 64, 2048);
        quick<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (U-Net dec 96->96 at 1012^2)", 96, 96, 1014);
        quick<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (96->96 at 253^2)", 96, 96, 255);
        quick<SplitCfg<3, 1, 128, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT128 4w (sub-pixel dec1.0: 104 -> 256 virtual at 1012^2)", 104, 256, 1014);
        quick<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2 (dec1.2 64->32 at 2024^2)", 64, 32, 2028);
        quick<SplitCfg<3, 1, 48, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT48 4w", 48, 48, 1024);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "r4a") {
        // round 4: wave tiles with more MFMAs per step for the narrow layers (NW = 8 pixel fragments per wave), the fcnn body
        quick<SplitCfg<11, 1, 64, 16, 64, 1, 8, 11, 1>, EPI_PLAIN>("K11 D1 MT64 8w 16x64 CC1 (fcnn body)", 64, 64, 2058);
        quick<SplitCfg<5, 4, 32, 16, 32, 2, 8, 5, 1>, EPI_PLAIN>("K5 D4 MT32 8w 16x32 CC2 (conv127, current)", 32, 32, 2064);
        quick<SplitCfg<5, 4, 32, 16, 64, 1, 8, 5, 1>, EPI_PLAIN>("K5 D4 MT32 8w 16x64 CC1 NW=8", 32, 32, 2064);
        quick<SplitCfg<5, 2, 32, 8, 32, 2, 4, 5, 1>, EPI_PLAIN>("K5 D2 MT32 4w 8x32 CC2 (conv127, current)", 32, 32, 2056);
        quick<SplitCfg<5, 2, 32, 16, 64, 1, 8, 5, 1>, EPI_PLAIN>("K5 D2 MT32 8w 16x64 CC1 NW=8", 32, 32, 2056);
        quick<SplitCfg<5, 2, 32, 8, 64, 1, 4, 5, 1>, EPI_PLAIN>("K5 D2 MT32 4w 8x64 CC1 NW=8", 32, 32, 2056);
        quick<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT64 4w 8x32 (ResNet8 conv0, current)", 64, 64, 2052);
        quick<SplitCfg<3, 2, 64, 16, 48, 2, 8, 3, 1>, EPI_PLAIN>("K3 D2 MT64 8w 16x48 NW=6", 64, 64, 2052);
        quick<SplitCfg<3, 2, 64, 16, 48, 2, 8, 3, 2>, EPI_PLAIN>("K3 D2 MT64 8w 16x48 NW=6 S=2", 64, 64, 2052);
        quick<SplitCfg<3, 2, 32, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT32 4w 8x32 (resnet8_u32, current)", 32, 32, 2052);
        quick<SplitCfg<3, 2, 32, 8, 64, 1, 4, 3, 1>, EPI_PLAIN>("K3 D2 MT32 4w 8x64 CC1 NW=8", 32, 32, 2052);
        quick<SplitCfg<3, 2, 32, 16, 64, 1, 8, 3, 1>, EPI_PLAIN>("K3 D2 MT32 8w 16x64 CC1 NW=8", 32, 32, 2052);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "r4b") {
        // 64-channel dilated layers: 16x48 tiles on 8 waves (NW = 6) against the current tiles, plain and residual epilogues
        quick<SplitCfg<3, 2, 64, 8, 32, 2, 4, 3, 1>, EPI_RES>("K3 D2 MT64 4w 8x32 RES (current)", 64, 64, 2052);
        quick<SplitCfg<3, 2, 64, 16, 48, 2, 8, 3, 1>, EPI_RES>("K3 D2 MT64 8w 16x48 RES", 64, 64, 2052);
        quick<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D4 MT64 8w 16x32 S=2 RES (current)", 64, 64, 2056);
        quick<SplitCfg<3, 4, 64, 16, 48, 2, 8, 3, 1>, EPI_RES>("K3 D4 MT64 8w 16x48 RES", 64, 64, 2056);
        quick<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D4 MT64 8w 16x32 S=2 (current)", 64, 64, 2056);
        quick<SplitCfg<3, 4, 64, 16, 48, 2, 8, 3, 1>, EPI_PLAIN>("K3 D4 MT64 8w 16x48", 64, 64, 2056);
        quick<SplitCfg<3, 8, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D8 MT64 8w 16x32 S=2 RES (current)", 64, 64, 2064);
        quick<SplitCfg<3, 1, 64, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT64 4w 8x32 (current)", 64, 64, 2050);
        quick<SplitCfg<3, 1, 64, 16, 48, 2, 8, 3, 1>, EPI_PLAIN>("K3 D1 MT64 8w 16x48", 64, 64, 2050);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "r4c") {
        // U-Net layers (full resolution / level 2): 8-wave tiles against the 4-wave ones
        quick<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w 8x32 (current) 96->96 at 1012^2", 96, 96, 1014);
        quick<SplitCfg<3, 1, 96, 16, 32, 2, 8, 3, 1>, EPI_PLAIN>("K3 D1 MT96 8w 16x32", 96, 96, 1014);
        quick<SplitCfg<3, 1, 96, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D1 MT96 8w 16x32 S=2", 96, 96, 1014);
        quick<SplitCfg<3, 1, 96, 16, 48, 2, 8, 3, 1>, EPI_PLAIN>("K3 D1 MT96 8w 16x48 NW=6", 96, 96, 1014);
        quick<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w 8x32 S=2 (current) dec1.2 64->32 at 2024^2", 64, 32, 2028);
        quick<SplitCfg<5, 1, 32, 16, 48, 2, 8, 5, 1>, EPI_PLAIN>("K5 D1 MT32 8w 16x48 NW=6", 64, 32, 2028);
        quick<SplitCfg<5, 1, 32, 16, 48, 2, 8, 5, 2>, EPI_PLAIN>("K5 D1 MT32 8w 16x48 NW=6 S=2", 64, 32, 2028);
        quick<SplitCfg<5, 1, 32, 16, 64, 1, 8, 5, 1>, EPI_PLAIN>("K5 D1 MT32 8w 16x64 CC1 NW=8", 64, 32, 2028);
        quick<SplitCfg<3, 1, 128, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT128 4w 8x32 (current) sub-pixel dec1.0 104->256v at 1012^2", 104, 256, 1014);
        quick<SplitCfg<3, 1, 128, 16, 32, 2, 8, 3, 1>, EPI_PLAIN>("K3 D1 MT128 8w 16x32", 104, 256, 1014);
        quick<SplitCfg<3, 1, 128, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D1 MT128 8w 16x32 S=2", 104, 256, 1014);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "unet") {
        bench<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w (U-Net dec 96->96 at 1012^2)", 96, 96, 1014);
        bench<SplitCfg<3, 1, 128, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT128 4w (sub-pixel dec1.0: 104 -> 256 virtual at 1012^2)", 104, 256, 1014);
        bench<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2 (dec1.2 64->32 at 2024^2)", 64, 32, 2028);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "stages") {
        // steps per stage (barrier every S steps), same layer, same process
        quick<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 1>, EPI_PLAIN>("K3 D2 MT64 8w S=1", 64, 64, 2048);
        quick<SplitCfg<3, 2, 64, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D2 MT64 8w S=2", 64, 64, 2048);
        quick<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 1>, EPI_RES>("K3 D4 MT64 8w RES S=1", 64, 64, 2048);
        quick<SplitCfg<3, 4, 64, 16, 32, 2, 8, 3, 2>, EPI_RES>("K3 D4 MT64 8w RES S=2", 64, 64, 2048);
        quick<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 1>, EPI_PLAIN>("K5 D1 MT32 4w S=1", 64, 32, 2048);
        quick<SplitCfg<5, 1, 32, 8, 32, 2, 4, 5, 2>, EPI_PLAIN>("K5 D1 MT32 4w S=2", 64, 32, 2048);
        quick<SplitCfg<3, 1, 96, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT96 4w S=1", 96, 96, 1024);
        quick<SplitCfg<3, 1, 48, 8, 32, 2, 4, 3, 1>, EPI_PLAIN>("K3 D1 MT48 4w S=1", 48, 48, 1024);
        quick<SplitCfg<3, 1, 48, 8, 32, 2, 4, 3, 2>, EPI_PLAIN>("K3 D1 MT48 4w S=2", 48, 48, 1024);
        quick<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 1>, EPI_PLAIN>("K3 D4 MT128 8w S=1", 128, 128, 2048);
        quick<SplitCfg<3, 4, 128, 16, 32, 2, 8, 3, 2>, EPI_PLAIN>("K3 D4 MT128 8w S=2", 128, 128, 2048);
        return 0;
    }
    bench<SplitCfg<3, 4, 128, 16, 32, 2>, EPI_RES>("K3 D4 MT128 RES (ResNet8 block2 conv1)", 64, 128, 2048);
    bench<SplitCfg<3, 4, 128, 16, 32, 2>, EPI_PLAIN>("K3 D4 MT128 PLAIN", 128, 128, 2048);
    bench<SplitCfg<5, 4, 128, 16, 32, 2>, EPI_HEAD>("K5 D4 MT128 HEAD", 128, 256, 2048);
    bench<SplitCfg<3, 2, 64, 16, 32, 2>, EPI_PLAIN>("K3 D2 MT64 PLAIN (ResNet8 conv0)", 64, 64, 2048);
    bench<SplitCfg<3, 8, 128, 16, 32, 2>, EPI_RES>("K3 D8 MT128 RES", 128, 128, 2048);
    return 0;
}
