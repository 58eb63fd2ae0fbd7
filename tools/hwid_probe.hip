#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, unsigned long long* t) {
    extern __shared__ float lds[];
    unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // all 32 bits of HW_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // burn some time so that two workgroups are co-resident
    float v = threadIdx.x;
    for (int i = 0; i < 20000; ++i) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id; t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t0; }
}
int main() {
    const int nb = 2048;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, nb * 4 * 4); hipMalloc(&dt, nb * 4 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 41 * 1024);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 41 * 1024, 0, d, dt);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4); std::vector<unsigned long long> ht(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), dt, nb * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, int> wave_hist, simd_hist;
    for (auto v : h) { wave_hist[v & 15]++; simd_hist[(v >> 4) & 3]++; }
    printf("wave_id histogram:"); for (auto& p : wave_hist) printf(" %u:%d", p.first, p.second); printf("\n");
    printf("simd_id histogram:"); for (auto& p : simd_hist) printf(" %u:%d", p.first, p.second); printf("\n");
    for (int b = 0; b < 6; ++b) printf("block %d: %08x %08x %08x %08x\n", b, h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    for (int b = 256; b < 260; ++b) printf("block %d: %08x %08x %08x %08x\n", b, h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    return 0;
}
