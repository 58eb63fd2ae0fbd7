// Sustained MFMA rate and shader clock of the f16 matrix instructions by shape (MI355X): does the 32x32x16 form, which reads
// half the operand registers per flop of the 16x16x32 form, hold a higher clock under the board's power management?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o tools/_run/mfma_clock && tools/_run/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <string>
#include <dirent.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void spin(const half8* __restrict__ src, float* __restrict__ out, int iters,
                                            unsigned long long* __restrict__ cyc) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 4095]; b[i] = src[(t * 8 + 4 + i) & 4095]; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float acc_out = 0.f;
    if (SHAPE == 0) {
        float4_ c[8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(a[u & 3]), "v"(b[(u + 1) & 3]));
        }
        for (int u = 0; u < 8; ++u) acc_out += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    } else if (SHAPE == 1) {
        float16_ c[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(a[u & 3]), "v"(b[(u + 1) & 3]));
        }
        for (int u = 0; u < 4; ++u)
            for (int k = 0; k < 16; ++k) acc_out += c[u][k];
    } else {
        float4_ c[8] = {};
        bf8 ab[4], bb[4];
        for (int i = 0; i < 4; ++i) { ab[i] = __builtin_bit_cast(bf8, a[i]); bb[i] = __builtin_bit_cast(bf8, b[i]); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[u]) : "v"(ab[u & 3]), "v"(bb[(u + 1) & 3]));
        }
        for (int u = 0; u < 8; ++u) acc_out += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    }
    const unsigned long long c1 = clock64();
    out[t] = acc_out;
    if (t == 0) { cyc[0] = c1 - c0; cyc[1] = wall_clock64() - w0; }
}

// board power: the largest power1_input of the DRM cards' hwmon nodes (one GPU is busy: it is that one), microwatts
static double read_power_w() {
    double best = 0;
    DIR* d = opendir("/sys/class/drm");
    if (!d) return 0;
    while (dirent* e = readdir(d)) {
        const std::string card = e->d_name;
        if (card.rfind("card", 0) != 0 || card.find('-') != std::string::npos) continue;
        const std::string hw = "/sys/class/drm/" + card + "/device/hwmon";
        DIR* h = opendir(hw.c_str());
        if (!h) continue;
        while (dirent* f = readdir(h)) {
            if (std::string(f->d_name).rfind("hwmon", 0) != 0) continue;
            FILE* fp = fopen((hw + "/" + f->d_name + "/power1_input").c_str(), "r");
            if (!fp) continue;
            double uw = 0;
            if (fscanf(fp, "%lf", &uw) == 1 && uw * 1e-6 > best) best = uw * 1e-6;
            fclose(fp);
        }
        closedir(h);
    }
    closedir(d);
    return best;
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && std::string(argv[1]) == "zeros";
    std::vector<_Float16> h(4096 * 8);
    srand(7);
    for (auto& v : h) v = zeros ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    printf("operands: %s\n", zeros ? "all zero" : "uniform in [-1, 1]");
    half8* d_src; float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_src, h.size() * 2); hipMalloc(&d_out, 4 * 256 * 2048); hipMalloc(&d_cyc, 16);
    hipMemcpy(d_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_bf16"};
    const int wgs = 256 * 2;                     // two 4-wave workgroups per CU: 2 waves per SIMD
    for (int rep = 0; rep < 2; ++rep)
        for (int shape = 0; shape < 3; ++shape) {
            const int iters = 1500000;           // ~0.4 s per launch; three launches back to back, the last one is reported
            float ms = 0;
            unsigned long long cyc[2] = {};
            std::atomic<bool> stop{false};
            double p_sum = 0; int p_n = 0;
            std::thread sampler([&] {
                while (!stop.load()) { const double p = read_power_w(); if (p > 0) { p_sum += p; ++p_n; } std::this_thread::sleep_for(std::chrono::milliseconds(20)); }
            });
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0);
                if (shape == 0) spin<0><<<wgs, 256>>>(d_src, d_out, iters, d_cyc);
                else if (shape == 1) spin<1><<<wgs, 256>>>(d_src, d_out, iters, d_cyc);
                else spin<2><<<wgs, 256>>>(d_src, d_out, iters, d_cyc);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            stop = true; sampler.join();
            hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost);
            const double flop_per_wave_iter = (shape == 1 ? 4 * 32.0 * 32 * 16 * 2 : 8 * 16.0 * 16 * 32 * 2);
            const double flops = flop_per_wave_iter * iters * (double)wgs * 4;
            // s_memtime ticks per s_memrealtime tick (100 MHz): the clock the counter follows, the nominal 2.4 GHz when the
            // operands are zero; `issue` = the rate over what that clock allows (1024 flop per SIMD and cycle)
            const double ghz = cyc[1] ? (double)cyc[0] / cyc[1] * 0.1 : 0.0;
            const double tf = flops / ms * 1e-9;
            printf("%-26s %8.1f ms  %7.1f TFLOP/s   shader clock %5.3f GHz, issue %.3f, board power %4.0f W   (nominal: 2500 TFLOP/s at 2.4 GHz)\n",
                   names[shape], ms, tf, ghz, ghz > 0 ? tf / (2500.0 * ghz / 2.4) : 0.0, p_n ? p_sum / p_n : 0.0);
        }
    return 0;
}
