// The "split" activation format of the 2xf16 convolution path (conv_split.h).
//
// An fp32 value x is carried as two halves  hi = f16(x),  lo = f16(x - hi)  (x = hi + lo to 22 bits; products
// of halves are exact in the fp32 accumulator of v_mfma_f32_16x16x32_f16).  A tensor of C channels is stored as
// two planes (hi, then lo) of 16-byte cells:   plane[cell = c / 8][y][x] = 8 consecutive channels of one pixel.
// One cell is exactly one lane's B operand of the K = 32 MFMA and one 16-byte LDS-DMA granule.  The bytes per
// element (2 + 2) are those of fp32, so buffers are sized as for fp32 with C rounded up to a multiple of 8;
// the channels that pad the last cell are stored as zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpz {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

static constexpr float SPLIT_MAX = 65504.f;   // largest finite f16: beyond it the hi half is +-inf

__host__ __device__ inline size_t split_cells(int C) { return (size_t)(C + 7) / 8; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// four consecutive channels of one pixel -> 8 bytes of hi halves and 8 bytes of lo halves
__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo) {
    f16x4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h[r] = (_Float16)v[r];
        l[r] = (_Float16)(v[r] - (float)h[r]);
    }
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}
__device__ __forceinline__ void join4(uint2 hi, uint2 lo, float (&v)[4]) {
    const f16x4 h = __builtin_bit_cast(f16x4, hi), l = __builtin_bit_cast(f16x4, lo);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (float)h[r] + (float)l[r];
}
// the same on channel PAIRS (two-wide float vectors: v_pk_fma / v_pk_add / v_cvt_pk_f16_f32 on gfx950) -- the epilogue of
// conv_split_kernel is bound by its instruction count
__device__ __forceinline__ void split2(f32x2 v, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f32x2 join2(unsigned hi, unsigned lo) {
    return __builtin_convertvector(__builtin_bit_cast(f16x2, hi), f32x2) + __builtin_convertvector(__builtin_bit_cast(f16x2, lo), f32x2);
}
// ... with the mixed-precision FMAs of gfx950 (v_fma_mix_f32 / v_fma_mixlo_f16 / v_fma_mixhi_f16 read an f16 half of a register as
// an fp32 operand, the latter two round the fp32 result to f16 into one half of the destination): no v_cvt_f32_f16 in front of the
// arithmetic.  Bit-identical to split2 (v - (float)hi is exact in fp32, so there is one rounding, to f16, either way); checked
// against the conversions on the MI355X by tools/mix_probe.hip.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2m(f32x2 v, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(v[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(v[1]));
    lo = l;
}
// join2 with one v_fma_mix_f32 per value (both halves read as f16 operands: hi * 1.0 + lo): no conversions at all
__device__ __forceinline__ f32x2 join2m(unsigned hi, unsigned lo) {
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d0) : "v"(hi), "v"(lo));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d1) : "v"(hi), "v"(lo));
    return (f32x2){d0, d1};
}
// c + (float2)(the two f16 halves of h): two v_fma_mix_f32
__device__ __forceinline__ f32x2 add_halves(f32x2 c, unsigned h) {
    float d0, d1;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h), "v"(c[0]));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h), "v"(c[1]));
    return (f32x2){d0, d1};
}

}  // namespace tpz
