// Does an LDS-DMA buffer load (buffer_load_dwordx4 ... offen lds) write ZEROS for lanes whose offset lies outside the
// descriptor's range?  (The 2xf16 conv kernels could then drop their zero-block source and the divergent paths around
// out-of-image lanes.)   hipcc --offload-arch=gfx950 -O3 tools/bufdma_probe.hip -o tools/_bin/bufdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, unsigned nbytes, float* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* l = reinterpret_cast<float*>(lds);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = -7.f;      // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (mode == 1 && (threadIdx.x & 3) == 1) voff = 0xfffffff0u;           // far outside
    if (mode == 2 && (threadIdx.x & 3) == 1) voff = nbytes;                // just outside
    if (mode == 3 && (threadIdx.x & 3) == 1) voff = nbytes - 8;            // straddling the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = l[i];
}

int main() {
    float *src, *out;
    const unsigned n = 4096;
    hipMalloc(&src, n * 4 + 4096);
    hipMalloc(&out, 1024);
    std::vector<float> h(n + 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.f + (float)i;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, src, n * 4, out, mode);
        float r[256];
        hipMemcpy(r, out, 1024, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int lane = 0; lane < 6; ++lane) printf("  lane%d = %g %g %g %g", lane, r[lane * 4], r[lane * 4 + 1], r[lane * 4 + 2], r[lane * 4 + 3]);
        bool zeros = true, in_ok = true;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                const float v = r[lane * 4 + j];
                if (mode > 0 && (lane & 3) == 1) { if (mode < 3 && v != 0.f) zeros = false; }
                else if (v != 1.f + lane * 4 + j) in_ok = false;
            }
        printf("\n   -> in-range lanes %s, out-of-range lanes %s\n", in_ok ? "correct" : "WRONG", mode == 0 ? "-" : zeros ? "ZERO" : "not zero");
    }
    return 0;
}
