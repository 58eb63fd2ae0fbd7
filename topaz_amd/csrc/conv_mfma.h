// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces the torch.nn.Conv2d / Conv3d calls the reference issues on its hot path
// (topaz/model/features/resnet.py:129-133,294-302, topaz/model/features/basic.py:47-63,
//  topaz/denoising/models.py:79-128,457-512): dilated "valid" convs of the filled scoring
// nets and "same" convs of the U-Nets, with the bias / activation / residual / eval-BN
// epilogue fused (resnet.py:101-105,185-202).
//
// GEMM view: M = output channels, N = output pixels, K = Cin * taps.
//   one MFMA: A[16 co][4 k] * B[4 k][16 px] -> C[16 co][16 px], exact f32 (fmaf chain).
//   k-group of 4 = four consecutive input channels at one tap            (generic)
//                = four consecutive kx taps of the single input channel  (CIN1 stems)
// Workgroup = 256 threads = 4 waves; tile = MT output channels x (TD x TH x TW) pixels.
//   rows (and planes) of the tile are strided by the dilation D ("polyphase" in y/z), so
//   the LDS halo in y/z is K-1 rows instead of (K-1)*D; columns are contiguous with a
//   (K-1)*D halo so every global row segment is a coalesced read.
// LDS holds one channel chunk of the input tile plus the pre-packed weight chunk
// (host packs weights in the exact lane order of the A fragment, so the A read is
//  lds[step][mf][lane] and the B read is lds[ch][z][y][x] -- all offsets are immediates).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* in;          // [Cin][Din][Hin][Win]
    const float* in2;         // optional 2nd source: channels [Cin1, Cin) come from here (fused concat)
    const float* wpk;         // packed weights (see pack_weights)
    const float* bias;        // [Cout] or nullptr
    float* out;               // [Cout][Dout][Hout][Wout] (or nullptr when head is fused)
    const float* res;         // residual [Cout][Dres][Hres][Wres] or nullptr
    const float* post_scale;  // [Cout] affine applied after the residual add (eval BN), or nullptr
    const float* post_shift;
    const float* head_w;      // fused 1x1 head: [Cout] weights, or nullptr
    float* head_out;          // [Dout][Hout][Wout]
    float head_b;
    const float* nrm;         // device float[4] {in_scale, in_shift, out_scale, out_shift} or nullptr
    int norm_src;             // bit0: x' = x*in_scale+in_shift on in-bounds pixels of `in`; bit1: same for `in2`
    int norm_out;             // y' = y*out_scale+out_shift applied last
    int Cin, Cin1;            // Cin1 = channels taken from `in` (== Cin when no concat)
    int Din, Hin, Win;        // geometry of `in2`/logical input (after nearest upsample of `in`)
    int D1, H1, W1;           // geometry of `in` when it is nearest-upsampled to (Din,Hin,Win); else == Din..
    long long cs1, ps1; int pitch1;   // channel / plane / row strides (floats) of `in`  (views into larger images)
    long long cs2, ps2; int pitch2;   // ... of `in2`
    int cog_inner;            // co-groups looped inside the kernel (fused head), else 1
    int Cout, Dout, Hout, Wout;
    int pad;                  // zero padding on every side
    int Dres, Hres, Wres, res_crop;
    int n_chunks;
    float slope;              // activation: v > 0 ? v : v*slope   (1.0 = identity, 0.0 = ReLU)
    int tiles_x, tiles_y, tiles_z;
};

// PyTorch 'nearest' source index: min(floor(dst * (float)in/out), in-1)  
__device__ __forceinline__ int nearest_src(int dst, int in_sz, int out_sz) {
    if (in_sz == out_sz) return dst;
    float scale = (float)in_sz / (float)out_sz;
    int s = (int)floorf((float)dst * scale);
    return s < in_sz - 1 ? s : in_sz - 1;
}

template <int K_, int D_, int MT_, int TD_, int TH_, int TW_, int KG_, bool CIN1_, int DIMS_>
struct ConvCfg {
    static constexpr int K = K_, D = D_, MT = MT_, TD = TD_, TH = TH_, TW = TW_, KG = KG_, DIMS = DIMS_;
    static constexpr bool CIN1 = CIN1_;
    static constexpr int KZ = (DIMS == 3) ? K : 1;
    static constexpr int MW = MT / 16;
    static constexpr int ROWS = TD * TH;            // tile rows (z-major)
    static constexpr int RPW = ROWS / 4;            // rows per wave
    static constexpr int NFC = TW / 16;             // N fragments per row
    static constexpr int NW = RPW * NFC;
    static constexpr int KP = CIN1 ? ((K + 3) / 4 * 4) : K;   // kx taps padded to a k-group
    static constexpr int ITD = TD + KZ - 1;
    static constexpr int ITH = TH + K - 1;
    static constexpr int ITW = TW + (KP - 1) * D;
    static constexpr int RS = ITW;
    static constexpr int PS = ITH * RS;             // plane stride
    static constexpr int CS_RAW = ITD * PS;
    // channel stride == 16 (mod 32): the two 16-lane halves of a ds_read_b32 group hit disjoint banks
    static constexpr int CS = CIN1 ? CS_RAW : (((CS_RAW - 16 + 31) / 32) * 32 + 16);
    static constexpr int NCH = CIN1 ? 1 : 4 * KG;
    static constexpr int IN_FLOATS = ((NCH * CS + 3) / 4) * 4;
    static constexpr int NSTEP = CIN1 ? KZ * K * (KP / 4) : KG * KZ * K * K;
    static constexpr int W_FLOATS = NSTEP * MW * 64;
    static constexpr int LDS_BYTES = (IN_FLOATS + W_FLOATS) * 4;
    // a wave's RPW tile rows either sit inside one z-plane, or cover whole z-planes
    static constexpr bool IN_PLANE = (TH % RPW == 0);
    static_assert(IN_PLANE || (RPW % TH == 0), "wave rows must align with z-planes");
    static constexpr int row_off(int row, int kz, int ky) {
        return IN_PLANE ? kz * PS + (row + ky) * RS : (row / TH + kz) * PS + (row % TH + ky) * RS;
    }
    static_assert(ROWS % 4 == 0, "tile rows must split over 4 waves");
    static_assert(TW % 16 == 0 && MT % 16 == 0, "MFMA 16x16 fragments");
    static_assert(IN_FLOATS * 4 < 65536 && W_FLOATS * 4 <= 65536, "ds_read immediate offsets are 16 bit");
};

template <class C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int K = C::K, D = C::D, MW = C::MW, NW = C::NW, NFC = C::NFC, KZ = C::KZ;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_in = lds;
    float* lds_w = lds + C::IN_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- tile coordinates. y (and z) tiles are polyphase: row i of the tile is y0 + i*D.
    const int bx = blockIdx.x;
    int by = blockIdx.y;
    const int tyz = by;                       // by enumerates (z tile, y tile) x phases
    const int ty = tyz % a.tiles_y;
    const int tz = tyz / a.tiles_y;
    const int yb = ty / D, yph = ty % D;
    const int y0 = yb * (C::TH * D) + yph;
    int z0 = 0;
    if (C::DIMS == 3) { const int zb = tz / D, zph = tz % D; z0 = zb * (C::TD * D) + zph; }
    const int x0 = bx * C::TW;
    const int ybase = y0 - a.pad, xbase = x0 - a.pad, zbase = (C::DIMS == 3) ? z0 - a.pad : 0;

    // per-lane LDS read bases (floats)
    const float* bl = lds_in + (C::CIN1 ? (l4 * D + l15) : (l4 * C::CS + l15));
    const float* al = lds_w + lane;

    const bool ups = (a.H1 != a.Hin) || (a.W1 != a.Win) || (a.D1 != a.Din);
    float in_scale = 1.f, in_shift = 0.f, out_scale = 1.f, out_shift = 0.f;
    if (a.nrm) { in_scale = a.nrm[0]; in_shift = a.nrm[1]; out_scale = a.nrm[2]; out_shift = a.nrm[3]; }
    const bool norm1 = (a.norm_src & 1) != 0, norm2 = (a.norm_src & 2) != 0;

    float hsum[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) hsum[n] = 0.f;

  for (int cg = 0; cg < a.cog_inner; ++cg) {
    const int cog = blockIdx.z * a.cog_inner + cg;
    f32x4 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < a.n_chunks; ++ch) {
        __syncthreads();
        // ---- stage the input chunk: NCH channels x ITD x ITH x ITW, zero outside the image
        constexpr int TILE_ELEMS = C::ITD * C::ITH * C::ITW;
        constexpr int IN_ELEMS = C::NCH * TILE_ELEMS;
#pragma unroll 4
        for (int e = tid; e < IN_ELEMS; e += 256) {
            const int c = e / TILE_ELEMS;
            const int rem = e - c * TILE_ELEMS;
            const int zz = rem / (C::ITH * C::ITW);
            const int rem2 = rem - zz * (C::ITH * C::ITW);
            const int r = rem2 / C::ITW;
            const int x = rem2 - r * C::ITW;
            const int ci = ch * C::NCH + c;
            const int gy = ybase + r * D, gx = xbase + x;
            const int gz = (C::DIMS == 3) ? zbase + zz * D : 0;
            float v = 0.f;
            if (ci < a.Cin && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win &&
                (unsigned)gz < (unsigned)a.Din) {
                if (ci < a.Cin1) {
                    if (ups) {
                        const int sy = nearest_src(gy, a.H1, a.Hin), sx = nearest_src(gx, a.W1, a.Win);
                        const int sz = (C::DIMS == 3) ? nearest_src(gz, a.D1, a.Din) : 0;
                        v = a.in[(long long)ci * a.cs1 + (long long)sz * a.ps1 + (long long)sy * a.pitch1 + sx];
                    } else {
                        v = a.in[(long long)ci * a.cs1 + (long long)gz * a.ps1 + (long long)gy * a.pitch1 + gx];
                    }
                    if (norm1) v = v * in_scale + in_shift;
                } else {
                    v = a.in2[(long long)(ci - a.Cin1) * a.cs2 + (long long)gz * a.ps2 + (long long)gy * a.pitch2 + gx];
                    if (norm2) v = v * in_scale + in_shift;
                }
            }
            lds_in[c * C::CS + zz * C::PS + r * C::RS + x] = v;
        }
        // ---- stage the weight chunk (already in fragment order)
        {
            const float4* wsrc =
                reinterpret_cast<const float4*>(a.wpk + ((size_t)cog * a.n_chunks + ch) * C::W_FLOATS);
            float4* wdst = reinterpret_cast<float4*>(lds_w);
#pragma unroll 4
            for (int e = tid; e < C::W_FLOATS / 4; e += 256) wdst[e] = wsrc[e];
        }
        __syncthreads();

        // ---- MFMA over the chunk
        const float* blw = bl + (C::IN_PLANE ? ((wave * C::RPW) / C::TH) * C::PS + ((wave * C::RPW) % C::TH) * C::RS
                                             : wave * (C::RPW / C::TH) * C::PS);
        if constexpr (C::CIN1) {
#pragma unroll
            for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kg = 0; kg < C::KP / 4; ++kg) {
                    const int step = (kz * K + ky) * (C::KP / 4) + kg;
                    float av[MW], bv[NW];
#pragma unroll
                    for (int m = 0; m < MW; ++m) av[m] = al[(step * MW + m) * 64];
#pragma unroll
                    for (int n = 0; n < NW; ++n) {
                        const int row = n / NFC, cc = n % NFC;   // row within the wave
                        bv[n] = blw[C::row_off(row, kz, ky) + cc * 16 + kg * 4 * D];
                    }
#pragma unroll
                    for (int m = 0; m < MW; ++m)
#pragma unroll
                        for (int n = 0; n < NW; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kg = 0; kg < C::KG; ++kg)
#pragma unroll
            for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const int step = ((kg * KZ + kz) * K + ky) * K + kx;
                        float av[MW], bv[NW];
#pragma unroll
                        for (int m = 0; m < MW; ++m) av[m] = al[(step * MW + m) * 64];
#pragma unroll
                        for (int n = 0; n < NW; ++n) {
                            const int row = n / NFC, cc = n % NFC;
                            bv[n] = blw[kg * 4 * C::CS + C::row_off(row, kz, ky) + cc * 16 + kx * D];
                        }
#pragma unroll
                        for (int m = 0; m < MW; ++m)
#pragma unroll
                            for (int n = 0; n < NW; ++n)
                                acc[m][n] =
                                    __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
                    }
        }
    }

    // ---- epilogue: bias, residual, eval-BN affine, activation, (fused 1x1 head), store
    const size_t plane_out = (size_t)a.Hout * a.Wout;
    const size_t vol_out = plane_out * a.Dout;
    const size_t plane_res = (size_t)a.Hres * a.Wres;
    const size_t vol_res = plane_res * a.Dres;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int trow = wave * C::RPW + n / NFC;             // tile row, z-major
        const int ti_z = trow / C::TH, ti_y = trow % C::TH;
        const int oy = y0 + ti_y * D, oz = (C::DIMS == 3) ? z0 + ti_z * D : 0;
        const int ox = x0 + (n % NFC) * 16 + l15;
        const bool inb = (oy < a.Hout) && (ox < a.Wout) && (oz < a.Dout);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = cog * C::MT + m * 16 + l4 * 4 + r;
                if (co < a.Cout && inb) {
                    float v = acc[m][n][r];
                    if (a.bias) v += a.bias[co];
                    if (a.res) {
                        const int c = a.res_crop;
                        v += a.res[(size_t)co * vol_res + (size_t)(C::DIMS == 3 ? oz + c : 0) * plane_res +
                                   (size_t)(oy + c) * a.Wres + (ox + c)];
                    }
                    if (a.post_scale) v = v * a.post_scale[co] + a.post_shift[co];
                    v = v > 0.f ? v : v * a.slope;
                    if (a.head_w) {
                        hsum[n] += v * a.head_w[co];
                    } else {
                        if (a.norm_out) v = v * out_scale + out_shift;
                        a.out[(size_t)co * vol_out + (size_t)oz * plane_out + (size_t)oy * a.Wout + ox] = v;
                    }
                }
            }
        }
    }
  }  // cog loop

    if (a.head_w) {
        // fused 1x1 head: reduce over the four 16-lane groups (they hold different co of the same pixel)
        const size_t plane_o = (size_t)a.Hout * a.Wout;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int trow = wave * C::RPW + n / NFC;
            const int ti_z = trow / C::TH, ti_y = trow % C::TH;
            const int oy = y0 + ti_y * D, oz = (C::DIMS == 3) ? z0 + ti_z * D : 0;
            const int ox = x0 + (n % NFC) * 16 + l15;
            float h = hsum[n];
            h += __shfl_xor(h, 16, 64);
            h += __shfl_xor(h, 32, 64);
            if (l4 == 0 && oy < a.Hout && ox < a.Wout && oz < a.Dout) {
                h += a.head_b;
                if (a.norm_out) h = h * out_scale + out_shift;
                a.head_out[(size_t)oz * plane_o + (size_t)oy * a.Wout + ox] = h;
            }
        }
    }
}

}  // namespace tpz
