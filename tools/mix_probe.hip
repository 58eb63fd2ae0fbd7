// v_fma_mix_f32 / mixlo / mixhi against explicit conversions (split_fmt.h split2m / add_halves):
//   hipcc --offload-arch=gfx950 -O3 tools/mix_probe.hip -o tools/_bin/mix_probe && tools/_bin/mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int HI> __device__ __forceinline__ float mix_add(unsigned h, float c) {   // (float)half + c
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    else    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    return d;
}
template <int HI> __device__ __forceinline__ float mix_sub(float c, unsigned h) {   // c - (float)half
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    else    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned split_lo_mix(float v0, float v1, unsigned hi) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v1));
    return l;
}
__global__ void k2(const float* v, unsigned* out, int n) {        // hi / lo split of (v[i], v[i+1]): mixed FMAs vs conversions
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const f32x2 x = {v[i], v[i + 1] * 0.37f};
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f16x2 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x2), f16x2);
    const unsigned hi = __builtin_bit_cast(unsigned, h);
    out[2 * i] = __builtin_bit_cast(unsigned, l);
    out[2 * i + 1] = split_lo_mix(x[0], x[1], hi);
}
__global__ void k(const float* v, const unsigned* h, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f16x2 hh = __builtin_bit_cast(f16x2, h[i]);
    out[6 * i + 0] = mix_add<0>(h[i], v[i]);
    out[6 * i + 1] = mix_add<1>(h[i], v[i]);
    out[6 * i + 2] = (float)hh[0] + v[i];
    out[6 * i + 3] = (float)hh[1] + v[i];
    out[6 * i + 4] = mix_sub<0>(v[i], h[i]) - (v[i] - (float)hh[0]);
    out[6 * i + 5] = mix_sub<1>(v[i], h[i]) - (v[i] - (float)hh[1]);
}
__device__ __forceinline__ f32x2 join2m(unsigned hi, unsigned lo) {      // split_fmt.h: hi * 1.0 + lo, both read as f16 halves
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d0) : "v"(hi), "v"(lo));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d1) : "v"(hi), "v"(lo));
    return (f32x2){d0, d1};
}
__global__ void k3(const unsigned* h, unsigned* bad, int n) {      // join of (hi, lo) pairs: one mixed FMA vs two conversions + add
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    const unsigned hi = h[i], lo = (h[i + 1] & 0x83ff83ffu) | 0x10001000u;      // lo halves ~2^-11 of the hi halves' range, some subnormal-ish
    const f32x2 a = join2m(hi, lo);
    const f32x2 b = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2) + __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
    const f32x2 c = join2m(hi, h[i + 1] & 0x03ff03ffu);                         // subnormal lo halves
    const f32x2 d = __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2) +
                    __builtin_convertvector(__builtin_bit_cast(f16x2, h[i + 1] & 0x03ff03ffu), f32x2);
    if (a[0] != b[0] || a[1] != b[1] || c[0] != d[0] || c[1] != d[1]) atomicAdd(bad, 1u);
}
int main() {
    const int n = 1 << 16;
    std::vector<float> v(n); std::vector<unsigned> h(n);
    uint32_t s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = ((int)(s >> 8) - (1 << 23)) * 1e-5f; s = s * 1664525u + 1013904223u;
        unsigned a = 0x2800 + (s >> 8) % 0x3000, b = 0x2800 + (s >> 20) % 0x3000 | ((s & 1) << 15); h[i] = a | b << 16; }
    float *dv, *dout; unsigned* dh;
    hipMalloc(&dv, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dout, n * 24);
    hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dh, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dv, dh, dout, n);
    std::vector<float> o(6 * n);
    hipMemcpy(o.data(), dout, n * 24, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) { if (o[6*i] != o[6*i+2] || o[6*i+1] != o[6*i+3] || o[6*i+4] != 0.f || o[6*i+5] != 0.f) ++bad; }
    printf("v_fma_mix_f32 probe: %d of %d mismatches\n", bad, n);
    unsigned* d2; hipMalloc(&d2, n * 8); hipMemset(d2, 0, n * 8);
    hipLaunchKernelGGL(k2, dim3(n / 256), dim3(256), 0, 0, dv, d2, n);
    std::vector<unsigned> o2(2 * n);
    hipMemcpy(o2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    int bad2 = 0;
    for (int i = 0; i + 1 < n; ++i) if (o2[2 * i] != o2[2 * i + 1]) ++bad2;
    printf("v_fma_mixlo_f16 / mixhi_f16 split probe: %d of %d mismatches\n", bad2, n - 1);
    unsigned* d3; hipMalloc(&d3, 4); hipMemset(d3, 0, 4);
    hipLaunchKernelGGL(k3, dim3(n / 256), dim3(256), 0, 0, dh, d3, n);
    unsigned bad3 = 0;
    hipMemcpy(&bad3, d3, 4, hipMemcpyDeviceToHost);
    printf("v_fma_mix_f32 join (two f16 operands) probe: %u of %d mismatches\n", bad3, n - 1);
    return bad != 0 || bad2 != 0 || bad3 != 0;
}
