// HBM-bound helper kernels of the hot path: direct convolution for 1-output-channel layers
// and 1->1 filters, 2x max-pool, mean/std reductions, affine maps, crop/paste and 3-D tiling.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "conv_mfma.h"
#include "kernels_misc.h"
#include "split_fmt.h"

namespace tpz {

// ------------------------------------------------------------------------------------------
// Direct convolution, one thread per output pixel per output channel.  Used where M (= Cout)
// is 1 and the matrix cores have nothing to do: the U-Net output convs (denoising/models.py:127,
// 512), the 1->1 Gaussian / inverse-Gaussian / affine filters (filters.py:28-96).
// Weights are read through the scalar cache (uniform addresses), inputs through L1/L2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs a, const float* __restrict__ w, int K,
                                                          int KZ, int dil) {
    const int ox = a.wx0 + blockIdx.x * 64 + (threadIdx.x & 63);       // (launch window: ConvArgs::wy0 .. wx1)
    const int oy = a.wy0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    const int nz = a.wz1 - a.wz0;                                      // (planes of the launch window)
    const int oz = a.wz0 + (int)(blockIdx.z % nz);
    const int co = blockIdx.z / nz;
    if (ox >= a.wx1 || oy >= a.wy1) return;
    float out_scale = 1.f, out_shift = 0.f;
    if (a.nrm && a.norm_out) { out_scale = a.nrm[2]; out_shift = a.nrm[3]; }
    float acc = 0.f;
    const float* wc = w + (size_t)co * a.Cin * KZ * K * K;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* src = a.in + (long long)ci * a.cs1;
        for (int kz = 0; kz < KZ; ++kz) {
            const int gz = (KZ > 1) ? oz - a.pad + kz * dil : 0;
            if ((unsigned)gz >= (unsigned)a.Din) continue;
            for (int ky = 0; ky < K; ++ky) {
                const int gy = oy - a.pad + ky * dil;
                if ((unsigned)gy >= (unsigned)a.Hin) continue;
                const float* row = src + (long long)gz * a.ps1 + (long long)gy * a.pitch1;
                const float* wr = wc + ((size_t)(ci * KZ + kz) * K + ky) * K;
                for (int kx = 0; kx < K; ++kx) {
                    const int gx = ox - a.pad + kx * dil;
                    if ((unsigned)gx < (unsigned)a.Win) {
                        acc = fmaf(wr[kx], row[gx], acc);
                    }
                }
            }
        }
    }
    float v = acc;
    if (a.bias) v += a.bias[co];
    // residual [Cout][Dres][Hres][Wres], centre-cropped by res_crop (ResidA; UDenoiseNet3's x - dec1(h) with negated weights)
    if (a.res) v += a.res[(((size_t)co * a.Dres + oz + (a.Dres > 1 ? a.res_crop : 0)) * a.Hres + oy + a.res_crop) * a.Wres + ox + a.res_crop];
    v = v > 0.f ? v : v * a.slope;
    if (a.norm_out) v = v * out_scale + out_shift;
    a.out[((size_t)co * a.Dout + oz) * a.Hout * a.Wout + (size_t)oy * a.Wout + ox] = v;
}

// LDS-tiled variant for the U-Net output convs (Cin -> 1, k = 3 or 5, dilation 1; 2-D and 3-D): a workgroup
// computes a TD x TH x 64 pixel tile (1 x 16 x 64 in 2-D, 4 x 4 x 64 in 3-D), each thread a strip of 4 pixels;
// per input channel and tap row it reads 4+K-1 neighbouring values with two ds_read_b128 and issues 4*K FMAs
// with scalar (SGPR) weights.  HBM-bound: reads Cin floats and writes one per pixel (AI ~ 12 flop/B in 2-D).
template <int K, int DIMS>
__global__ __launch_bounds__(256) void conv_cout1_tiled_kernel(const ConvArgs a, const float* __restrict__ w) {
    constexpr int TD = DIMS == 3 ? 4 : 1, TH = DIMS == 3 ? 4 : 16, TW = 64;
    constexpr int KZ = DIMS == 3 ? K : 1;
    constexpr int CC = DIMS == 3 ? 4 : 8;
    constexpr int ITD = TD + KZ - 1, ITH = TH + K - 1, RS = (TW + K - 1 + 3) / 4 * 4 + 4, PS = ITH * RS, CS = ITD * PS;
    __shared__ __attribute__((aligned(16))) float tile[CC * CS];
    const int tid = threadIdx.x;
    const int sx = tid & 15, sy = (tid >> 4) % TH, sz = (tid >> 4) / TH;
    const int x0 = a.wx0 + blockIdx.x * TW, y0 = a.wy0 + blockIdx.y * TH, z0 = a.wz0 + blockIdx.z * TD;
    float out_scale = 1.f, out_shift = 0.f;
    if (a.nrm && a.norm_out) { out_scale = a.nrm[2]; out_shift = a.nrm[3]; }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < a.Cin; c0 += CC) {
        __syncthreads();
        for (int e = tid; e < CC * CS; e += 256) {
            const int c = e / CS, rem = e - c * CS, zz = rem / PS, rem2 = rem - zz * PS, r = rem2 / RS, x = rem2 - r * RS;
            const int gz = DIMS == 3 ? z0 - a.pad + zz : 0, gy = y0 - a.pad + r, gx = x0 - a.pad + x, ci = c0 + c;
            float v = 0.f;
            if (ci < a.Cin && (unsigned)gz < (unsigned)a.Din && (unsigned)gy < (unsigned)a.Hin &&
                (unsigned)gx < (unsigned)a.Win)
                v = a.in[(long long)ci * a.cs1 + (long long)gz * a.ps1 + (long long)gy * a.pitch1 + gx];
            tile[e] = v;
        }
        __syncthreads();
        const int nc = min(CC, a.Cin - c0);
        for (int c = 0; c < nc; ++c) {
            const float* wc = w + (size_t)(c0 + c) * KZ * K * K;      // uniform -> scalar loads
#pragma unroll
            for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const float4* row =
                    reinterpret_cast<const float4*>(tile + c * CS + (sz + kz) * PS + (sy + ky) * RS + sx * 4);
                const float4 v0 = row[0], v1 = row[1];
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float wv = wc[(kz * K + ky) * K + kx];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p] = fmaf(wv, v[p + kx], acc[p]);
                }
            }
        }
    }
    const int oy = y0 + sy, oz = z0 + sz;
    if (oy < a.wy1 && oz < a.wz1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ox = x0 + sx * 4 + p;
            if (ox < a.wx1) {
                float v = acc[p];
                if (a.bias) v += a.bias[0];
                if (a.res) v += a.res[(((size_t)oz + (a.Dres > 1 ? a.res_crop : 0)) * a.Hres + oy + a.res_crop) * a.Wres + ox + a.res_crop];
                v = v > 0.f ? v : v * a.slope;
                if (a.norm_out) v = v * out_scale + out_shift;
                a.out[((size_t)oz * a.Hout + oy) * a.Wout + ox] = v;
            }
        }
    }
}

hipError_t launch_conv_direct(const ConvArgs& a_in, const float* d_w, int K, int KZ, int dil, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.wy1 <= 0) { a.wy0 = a.wx0 = 0; a.wy1 = a.Hout; a.wx1 = a.Wout; }      // no window: the whole tensor
    if (a.wz1 <= 0 || a.Dout <= 1) { a.wz0 = 0; a.wz1 = a.Dout > 1 ? a.Dout : 1; }
    const int wh = a.wy1 - a.wy0, ww = a.wx1 - a.wx0, wd = a.wz1 - a.wz0;
    if (KZ == 1 && dil == 1 && a.Cout == 1 && (K == 3 || K == 5)) {
        dim3 grid((ww + 63) / 64, (wh + 15) / 16, 1);
        if (K == 3) hipLaunchKernelGGL((conv_cout1_tiled_kernel<3, 2>), grid, dim3(256), 0, s, a, d_w);
        else hipLaunchKernelGGL((conv_cout1_tiled_kernel<5, 2>), grid, dim3(256), 0, s, a, d_w);
        return hipGetLastError();
    }
    if (KZ == 3 && K == 3 && dil == 1 && a.Cout == 1 && (wd + 3) / 4 <= 65535) {
        dim3 grid((ww + 63) / 64, (wh + 3) / 4, (wd + 3) / 4);
        hipLaunchKernelGGL((conv_cout1_tiled_kernel<3, 3>), grid, dim3(256), 0, s, a, d_w);
        return hipGetLastError();
    }
    dim3 grid((ww + 63) / 64, (wh + 3) / 4, wd * a.Cout);
    hipLaunchKernelGGL(conv_direct_kernel, grid, dim3(256), 0, s, a, d_w, K, KZ, dil);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(2) / MaxPool3d(2), floor mode (denoising/models.py:81-97,459-479)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                       int D, int H, int W, int Do, int Ho, int Wo, int dims) {
    const size_t n = (size_t)C * Do * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        size_t t = i / Wo;
        const int y = (int)(t % Ho);
        t /= Ho;
        const int z = (int)(t % Do);
        const int c = (int)(t / Do);
        const float* p = in + (((size_t)c * D + (dims == 3 ? 2 * z : 0)) * H + 2 * y) * W + 2 * x;
        float m = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
        if (dims == 3) {
            const float* q = p + (size_t)H * W;
            m = fmaxf(m, fmaxf(fmaxf(q[0], q[1]), fmaxf(q[W], q[W + 1])));
        }
        out[i] = m;
    }
}

hipError_t launch_maxpool2(const float* in, float* out, int C, int D, int H, int W, int dims, hipStream_t s) {
    const int Do = dims == 3 ? D / 2 : 1, Ho = H / 2, Wo = W / 2;
    const size_t n = (size_t)C * Do * Ho * Wo;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(blocks), dim3(256), 0, s, in, out, C, D, H, W, Do, Ho, Wo, dims);
    return hipGetLastError();
}

// MaxPool2d(2) / MaxPool3d(2) on split cells: the (hi, lo) pair of the largest of the 4 / 8 values is selected
// per channel, so the result equals the pooled fp32 value exactly.
__global__ __launch_bounds__(256) void maxpool2_split_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                             int cells, int D, int H, int W, int Do, int Ho, int Wo,
                                                             int dims) {
    const size_t n = (size_t)cells * Do * Ho * Wo;
    const size_t plane_in = (size_t)cells * D * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        size_t t = i / Wo;
        const int y = (int)(t % Ho);
        t /= Ho;
        const int z = (int)(t % Do);
        const size_t c = t / Do;
        const size_t p = ((c * D + (dims == 3 ? 2 * z : 0)) * H + 2 * y) * W + 2 * x;
        f16x8 bh = __builtin_bit_cast(f16x8, in[p]), bl = __builtin_bit_cast(f16x8, in[plane_in + p]);
        const int nz = dims == 3 ? 2 : 1;
        for (int k = 1; k < 4 * nz; ++k) {
            const size_t q = p + (size_t)(k >> 2) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1);
            const f16x8 h = __builtin_bit_cast(f16x8, in[q]), l = __builtin_bit_cast(f16x8, in[plane_in + q]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // hi + lo compares as the fp32 value it stands for
                const float a = (float)bh[j] + (float)bl[j], b = (float)h[j] + (float)l[j];
                if (b > a || a != a) { bh[j] = h[j]; bl[j] = l[j]; }   // fmaxf semantics, as maxpool2_kernel
            }
        }
        out[i] = __builtin_bit_cast(uint4, bh);
        out[n + i] = __builtin_bit_cast(uint4, bl);
    }
}

// The z half of MaxPool3d(2) on split cells: out[c][z] = max(in[c][2z], in[c][2z + 1]) plane-wise.  The in-plane half ran in
// the producing conv's epilogue (EPI_POOL on the plane-stacked kernels, which see one output plane at a time): the conv then
// writes a quarter of its tensor and this kernel reads that quarter instead of all of it.
__global__ __launch_bounds__(256) void maxpoolz_split_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int cells,
                                                             int D, int Do, size_t hw) {
    const size_t n = (size_t)cells * Do * hw, plane_in = (size_t)cells * D * hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t px = i % hw, t = i / hw;
        const int z = (int)(t % Do);
        const size_t c = t / Do;
        const size_t p = (c * D + 2 * z) * hw + px, q = p + hw;
        f16x8 bh = __builtin_bit_cast(f16x8, in[p]), bl = __builtin_bit_cast(f16x8, in[plane_in + p]);
        const f16x8 h = __builtin_bit_cast(f16x8, in[q]), l = __builtin_bit_cast(f16x8, in[plane_in + q]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)bh[j] + (float)bl[j], b = (float)h[j] + (float)l[j];
            if (b > a || a != a) { bh[j] = h[j]; bl[j] = l[j]; }       // (as maxpool2_split_kernel)
        }
        out[i] = __builtin_bit_cast(uint4, bh);
        out[n + i] = __builtin_bit_cast(uint4, bl);
    }
}

hipError_t launch_maxpoolz_split(const float* in, float* out, int C, int D, int H, int W, hipStream_t s) {
    const int Do = D / 2;
    const size_t n = split_cells(C) * (size_t)Do * H * W;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpoolz_split_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)in, (uint4*)out, (int)split_cells(C), D,
                       Do, (size_t)H * W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Dilated k^dims max over a window, stride 1, no padding: the FILLED form of MaxPool(3, stride = 2) in the ResNets trained
// with --pooling max and in ResNet6 (resnet.py:10-47: fill() turns the stride into a dilation of everything downstream and
// the pool itself into a stride-1 pool at the accumulated dilation).  fp32 planes and split cells.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpoolk_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int D,
                                                       int H, int W, int Do, int Ho, int Wo, int k, int dil, int dims) {
    const size_t n = (size_t)C * Do * Ho * Wo;
    const int kz_n = dims == 3 ? k : 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        size_t t = i / Wo;
        const int y = (int)(t % Ho);
        t /= Ho;
        const int z = (int)(t % Do);
        const size_t c = t / Do;
        const float* p = in + ((c * D + z) * H + y) * W + x;
        float m = p[0];
        for (int kz = 0; kz < kz_n; ++kz)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const float v = p[((size_t)kz * dil * H + (size_t)ky * dil) * W + (size_t)kx * dil];
                    m = (v > m || v != v) ? v : m;            // a NaN propagates, as in torch's max_pool
                }
        out[i] = m;
    }
}
__global__ __launch_bounds__(256) void maxpoolk_split_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int cells,
                                                             int D, int H, int W, int Do, int Ho, int Wo, int k, int dil,
                                                             int dims) {
    const size_t n = (size_t)cells * Do * Ho * Wo;
    const size_t plane_in = (size_t)cells * D * H * W;
    const int kz_n = dims == 3 ? k : 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        size_t t = i / Wo;
        const int y = (int)(t % Ho);
        t /= Ho;
        const int z = (int)(t % Do);
        const size_t c = t / Do;
        const size_t p = ((c * D + z) * H + y) * W + x;
        f16x8 bh = __builtin_bit_cast(f16x8, in[p]), bl = __builtin_bit_cast(f16x8, in[plane_in + p]);
        for (int kz = 0; kz < kz_n; ++kz)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx) {
                    const size_t q = p + ((size_t)kz * dil * H + (size_t)ky * dil) * W + (size_t)kx * dil;
                    const f16x8 h = __builtin_bit_cast(f16x8, in[q]), l = __builtin_bit_cast(f16x8, in[plane_in + q]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float a = (float)bh[j] + (float)bl[j], b = (float)h[j] + (float)l[j];   // the fp32 values they stand for
                        if (b > a || b != b) { bh[j] = h[j]; bl[j] = l[j]; }
                    }
                }
        out[i] = __builtin_bit_cast(uint4, bh);
        out[n + i] = __builtin_bit_cast(uint4, bl);
    }
}
hipError_t launch_maxpoolk(const void* in, void* out, int C, int D, int H, int W, int k, int dil, int dims, bool split,
                           hipStream_t s) {
    const int span = dil * (k - 1);
    const int Do = dims == 3 ? D - span : 1, Ho = H - span, Wo = W - span;
    const size_t cc = split ? split_cells(C) : (size_t)C;
    const size_t n = cc * Do * Ho * Wo;
    if (n == 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
    if (split)
        hipLaunchKernelGGL(maxpoolk_split_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)in, (uint4*)out, (int)cc, D, H, W,
                           Do, Ho, Wo, k, dil, dims);
    else
        hipLaunchKernelGGL(maxpoolk_kernel, dim3(blocks), dim3(256), 0, s, (const float*)in, (float*)out, C, D, H, W, Do, Ho, Wo,
                           k, dil, dims);
    return hipGetLastError();
}

hipError_t launch_maxpool2_split(const void* in, void* out, int C, int D, int H, int W, int dims, hipStream_t s) {
    const int Do = dims == 3 ? D / 2 : 1, Ho = H / 2, Wo = W / 2;
    const size_t n = split_cells(C) * (size_t)Do * Ho * Wo;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool2_split_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)in, (uint4*)out,
                       (int)split_cells(C), dims == 3 ? D : 1, H, W, Do, Ho, Wo, dims);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// mean / std of a (strided) 3-D box, deterministic two-stage reduction in fp64.
// Stage 1 writes per-block partial (sum, sumsq); stage 2 (one block) folds them in fixed order
// and emits {mean, std} or directly the normalisation parameters the conv kernels consume.
// torch.std is unbiased (N-1), numpy.std is population (N)  (denoise.py:283 vs :343,:388).
// sum of squares about a pilot value (the first element) keeps the one-pass variance stable.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void meanstd_partial_kernel(const float* __restrict__ x, int D, int H, int W,
                                                              long long ps, int pitch, double* __restrict__ part) {
    __shared__ double s1[4], s2[4];
    const size_t n = (size_t)D * H * W;
    const float pilot = x[0];
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const size_t t = i / W;
        const int yy = (int)(t % H);
        const int zz = (int)(t / H);
        const double v = (double)x[(long long)zz * ps + (long long)yy * pitch + xx] - (double)pilot;
        a += v;
        b += v * v;
    }
    a = wave_sum(a);
    b = wave_sum(b);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s1[wv] = a; s2[wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s1[0] + s1[1] + s1[2] + s1[3];
        part[2 * blockIdx.x + 1] = s2[0] + s2[1] + s2[2] + s2[3];
    }
}

// mode 0: out = {mean, std}
// mode 1: out = {1/std, -mean/std, std, mean}                        (Denoise._denoise, denoise.py:283-295)
// mode 2: out = {1/std, -mean/std, std*gstd, mean*gstd + gmean}      (3-D tiles: ..., then *std+mu of the volume, :355)
__global__ __launch_bounds__(256) void meanstd_final_kernel(const double* __restrict__ part, int nblocks,
                                                            const float* __restrict__ x0, double n, int unbiased,
                                                            int mode, const float* __restrict__ g, float* __restrict__ out) {
    __shared__ double s1[4], s2[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = wave_sum(a);
    b = wave_sum(b);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s1[wv] = a; s2[wv] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sa = s1[0] + s1[1] + s1[2] + s1[3];
        const double sb = s2[0] + s2[1] + s2[2] + s2[3];
        const double dm = sa / n;
        const double mean = dm + (double)x0[0];
        double var = (sb - sa * dm) / (unbiased ? (n - 1.0) : n);
        if (var < 0.0) var = 0.0;
        const float mu = (float)mean, sd = (float)sqrt(var);
        if (mode == 0) {
            out[0] = mu; out[1] = sd;
        } else {
            // the reference computes (x - mu)/std and pred*std + mu in fp32 (denoise.py:284,295)
            out[0] = 1.0f / sd;
            out[1] = -mu / sd;
            if (mode == 1) { out[2] = sd; out[3] = mu; }
            else { out[2] = sd * g[1]; out[3] = mu * g[1] + g[0]; }
        }
    }
}

hipError_t launch_meanstd(const float* x, int D, int H, int W, long long ps, int pitch, int unbiased, int mode,
                          const float* d_g, double* d_part, int part_blocks, float* d_out, hipStream_t s) {
    const size_t n = (size_t)D * H * W;
    int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    if (blocks < 1) blocks = 1;
    if (blocks > part_blocks) blocks = part_blocks;
    hipLaunchKernelGGL(meanstd_partial_kernel, dim3(blocks), dim3(256), 0, s, x, D, H, W, ps, pitch, d_part);
    hipLaunchKernelGGL(meanstd_final_kernel, dim3(1), dim3(256), 0, s, d_part, blocks, x, (double)n, unbiased, mode,
                       d_g, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// elementwise affine y = x*scale + shift (host scalars) or with device params p[si], p[hi]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                     float scale, float shift) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = x[i] * scale + shift;
}

hipError_t launch_affine(const float* x, float* y, size_t n, float scale, float shift, hipStream_t s) {
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(affine_kernel, dim3(blocks), dim3(256), 0, s, x, y, n, scale, shift);
    return hipGetLastError();
}

// y = (x - mu) / std in fp32, as numpy evaluates `(mic - mu) / std` (denoise.py:389): the subtraction first -- a raw-count
// micrograph with |mu| >> std keeps its low bits, which x * (1 / std) + (-mu / std) with two rounded coefficients does not --
// and an IEEE division (std == 0 gives the reference's inf / nan, not a host exception)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float mu,
                                                        float sd) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        y[i] = __fdiv_rn(__fsub_rn(x[i], mu), sd);
}

hipError_t launch_normalize(const float* x, float* y, size_t n, float mu, float sd, hipStream_t s) {
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(normalize_kernel, dim3(blocks), dim3(256), 0, s, x, y, n, mu, sd);
    return hipGetLastError();
}

// y[z][y][x] (dense) = x_view[z][y][x]*p[0] + p[1] with p on the device: the (x - mu)/std step of
// Denoise._denoise (denoise.py:284) applied to a (strided) patch view before the network reads it.
__global__ __launch_bounds__(256) void affine_dev_kernel(const float* __restrict__ x, int D, int H, int W,
                                                         long long ps, int pitch, const float* __restrict__ p,
                                                         float* __restrict__ y) {
    const size_t n = (size_t)D * H * W;
    const float sc = p[0], sh = p[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const size_t t = i / W;
        const int yy = (int)(t % H);
        const int zz = (int)(t / H);
        y[i] = x[(long long)zz * ps + (long long)yy * pitch + xx] * sc + sh;
    }
}

hipError_t launch_affine_dev(const float* x, int D, int H, int W, long long ps, int pitch, const float* d_p, float* y,
                             hipStream_t s) {
    const size_t n = (size_t)D * H * W;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(affine_dev_kernel, dim3(blocks), dim3(256), 0, s, x, D, H, W, ps, pitch, d_p, y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Range scaling of a scoring pass.  `topaz extract` scores micrographs as they come (extract.py:234-249: no normalisation), and a
// raw-count image (mean 10^3 .. 10^4) drives the activations of the 2xf16 path past the f16 range.  A scoring network is
// positively homogeneous in (input, biases) jointly -- convolutions, PReLU / ReLU, max-pools, residual adds, eval-BN affines and
// the linear head are -- so f(2^-s x; 2^-s b) = 2^-s f(x; b), exactly (powers of two): the pass runs on x * 2^-s with every
// bias-like vector scaled alike and multiplies its result by 2^s.  s is chosen per image on the device (no host round trip).
// ------------------------------------------------------------------------------------------
// histogram of the biased exponents of |x| (integer counts: deterministic whatever the order of the atomics)
__global__ __launch_bounds__(256) void exp_hist_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        atomicAdd(&h[(__float_as_uint(x[i]) >> 23) & 0xffu], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// s from the histogram: the exponent e below which all but n / 1024 of the finite pixels lie (the 99.9 % quantile of |x|, to
// a power of two) is brought to 2^3: s = max(0, e - 127 - 3).  The BULK of the image decides, not its maximum: a hot pixel a
// thousand times the rest would otherwise push every other pixel towards the f16 subnormals (the lo halves first) without
// tripping any flag; here it either still fits the f16 range after scaling or trips the overflow flag and the image is
// re-run on the fp32 kernels -- exact either way.  N(0,1) images: e - 127 = 1, s = 0 (untouched).
__global__ __launch_bounds__(256) void range_finish_kernel(unsigned* __restrict__ hist, size_t n, float* __restrict__ rng,
                                                           const float* __restrict__ src, float* __restrict__ dst, size_t n_vec) {
    __shared__ int s_sh;
    if (threadIdx.x == 0) {
        const size_t finite = n - hist[255];
        const size_t allow = finite >> 10;
        size_t above = 0;
        int e = 254;
        for (; e > 0; --e) {
            above += hist[e];
            if (above > allow) break;
        }
        int s = e - 127 - 3;
        s = s < 0 ? 0 : s > 120 ? 120 : s;
        if (finite == 0) s = 0;
        s_sh = s;
    }
    __syncthreads();
    const int s = s_sh;
    const float down = ldexpf(1.f, -s), up = ldexpf(1.f, s);
    for (size_t i = threadIdx.x; i < n_vec; i += 256) dst[i] = src[i] * down;
    __syncthreads();
    hist[threadIdx.x] = 0;                          // ready for the next image on this stream
    if (threadIdx.x == 0) { rng[0] = down; rng[1] = 0.f; rng[2] = up; rng[3] = (float)s; }
}

hipError_t launch_range_fit(const float* x, size_t n, unsigned* hist, float* rng, const float* src, float* dst, size_t n_vec,
                            hipStream_t s) {
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    if (n) hipLaunchKernelGGL(exp_hist_kernel, dim3(blocks), dim3(256), 0, s, x, n, hist);
    hipLaunchKernelGGL(range_finish_kernel, dim3(1), dim3(256), 0, s, hist, n, rng, src, dst, n_vec);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void unscale_kernel(float* __restrict__ y, size_t n, const float* __restrict__ rng, float add) {
    const float up = rng[2];
    if (up == 1.f && add == 0.f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = y[i] * up + add;
}

hipError_t launch_unscale(float* y, size_t n, const float* rng, float add, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(unscale_kernel, dim3(blocks), dim3(256), 0, s, y, n, rng, add);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// copy a box: dst[dz0+z][dy0+y][dx0+x] = src[sz0+z][sy0+y][sx0+x]   (stitching, denoise.py:322,365-369)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_box_kernel(const float* __restrict__ src, long long sps, int spitch,
                                                       float* __restrict__ dst, long long dps, int dpitch, int bd,
                                                       int bh, int bw) {
    const size_t n = (size_t)bd * bh * bw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % bw);
        const size_t t = i / bw;
        const int y = (int)(t % bh);
        const int z = (int)(t / bh);
        dst[(long long)z * dps + (long long)y * dpitch + x] = src[(long long)z * sps + (long long)y * spitch + x];
    }
}

hipError_t launch_copy_box(const float* src, long long sps, int spitch, float* dst, long long dps, int dpitch, int bd,
                           int bh, int bw, hipStream_t s) {
    const size_t n = (size_t)bd * bh * bw;
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(copy_box_kernel, dim3(blocks), dim3(256), 0, s, src, sps, spitch, dst, dps, dpitch, bd, bh, bw);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// fp32 [C][H][W]  <->  split cells (split_fmt.h).  Interop / test helpers of the 2xf16 path: inside a
// network the conversions are fused into the producing kernels' epilogues.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_split_kernel(const float* __restrict__ in, uint4* __restrict__ out, int C,
                                                       size_t hw, unsigned* flag) {
    bool big = false;
    const size_t cells = split_cells(C);
    const size_t n = cells * hw;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t cell = i / hw, px = i - cell * hw;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t c = cell * 8 + j;
            v[j] = c < (size_t)C ? in[c * hw + px] : 0.f;
            big |= !(fabsf(v[j]) <= SPLIT_MAX);
        }
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        uint2 h0, l0, h1, l1;
        split4(a, h0, l0);
        split4(b, h1, l1);
        out[i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[n + i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    if (big && flag) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(256) void from_split_kernel(const uint4* __restrict__ in, float* __restrict__ out, int C,
                                                         size_t hw) {
    const size_t cells = split_cells(C);
    const size_t n = cells * hw;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t cell = i / hw, px = i - cell * hw;
        const uint4 h = in[i], l = in[n + i];
        float a[4], b[4];
        join4(make_uint2(h.x, h.y), make_uint2(l.x, l.y), a);
        join4(make_uint2(h.z, h.w), make_uint2(l.z, l.w), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t c = cell * 8 + j;
            if (c < (size_t)C) out[c * hw + px] = j < 4 ? a[j] : b[j - 4];
        }
    }
}

hipError_t launch_to_split(const float* in, void* out, int C, int H, int W, unsigned* flag, hipStream_t s) {
    const size_t n = split_cells(C) * (size_t)H * W;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(to_split_kernel, dim3(blocks), dim3(256), 0, s, in, (uint4*)out, C, (size_t)H * W, flag);
    return hipGetLastError();
}
hipError_t launch_from_split(const void* in, float* out, int C, int H, int W, hipStream_t s) {
    const size_t n = split_cells(C) * (size_t)H * W;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(from_split_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)in, out, C, (size_t)H * W);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// One pass of the 2-component Gaussian-mixture fit behind `topaz normalize` (topaz/stats.py:120-203 gmm_fit).
// mode 0: hard assignment p1 = (x > split) (the initialisation, stats.py:131-134);
// mode 1: E-step with the current parameters: log p_k = -(x-mu_k)^2/(2 var_k) - ln(2 pi var_k)/2 + ln(pi_k),
//         Z = logsumexp, p_k = exp(log p_k - Z).
// Accumulates, in fp64 and in a fixed order (deterministic), S = {sum Z, sum p0, sum p1, sum p0 x, sum p1 x,
// sum p0 x^2, sum p1 x^2}: the M-step and the log-likelihood are closed forms of S (rt_stats.hip gmm_fit).
// par = {split | mu0, mu1, var0, var1, ln(1-pi), ln(pi)}.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gmm_pass_kernel(const float* __restrict__ x, size_t n, int mode,
                                                       const double* __restrict__ par, double* __restrict__ part) {
    __shared__ double sh[4][7];
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const double p0_ = par[0], p1_ = par[1], p2_ = par[2], p3_ = par[3], p4_ = par[4], p5_ = par[5];
    const double c0 = mode ? -0.5 * log(2.0 * 3.14159265358979323846 * p2_) + p4_ : 0.0;
    const double c1 = mode ? -0.5 * log(2.0 * 3.14159265358979323846 * p3_) + p5_ : 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = (double)x[i];
        double q0, q1, Z = 0.0;
        if (mode == 0) {
            q0 = v <= p0_ ? 1.0 : 0.0;
            q1 = 1.0 - q0;
        } else {
            const double l0 = -(v - p0_) * (v - p0_) / 2.0 / p2_ + c0;
            const double l1 = -(v - p1_) * (v - p1_) / 2.0 / p3_ + c1;
            const double ma = l0 > l1 ? l0 : l1;
            Z = ma + log(exp(l0 - ma) + exp(l1 - ma));
            q0 = exp(l0 - Z);
            q1 = exp(l1 - Z);
        }
        acc[0] += Z; acc[1] += q0; acc[2] += q1;
        acc[3] += q0 * v; acc[4] += q1 * v;
        acc[5] += q0 * v * v; acc[6] += q1 * v * v;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = wave_sum(acc[k]);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 7; ++k) sh[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 7) part[(size_t)blockIdx.x * 7 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void gmm_final_kernel(const double* __restrict__ part, int blocks, double* __restrict__ out) {
    if (threadIdx.x < 7) {
        double s = 0.0;
        for (int b = 0; b < blocks; ++b) s += part[(size_t)b * 7 + threadIdx.x];
        out[threadIdx.x] = s;
    }
}

hipError_t launch_gmm_pass(const float* x, size_t n, int mode, const double* d_par, double* d_part, int part_blocks,
                           double* d_out, hipStream_t s) {
    int blocks = (int)((n + 255) / 256 < (size_t)part_blocks ? (n + 255) / 256 : (size_t)part_blocks);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gmm_pass_kernel, dim3(blocks), dim3(256), 0, s, x, n, mode, d_par, d_part);
    hipLaunchKernelGGL(gmm_final_kernel, dim3(1), dim3(64), 0, s, d_part, blocks, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Column-kernel helpers of the 2xf16 path (rt_load.hip prepare_split).
// shiftx_split: out cell jc of pixel (z, y, x) = the 8 values x[z][y][x - pad + 8*jc + j], j = 0..7 (0 outside the
// image or for taps >= K) as split f16 halves: the kx taps of a 1-channel stem become input channels.
// shiftsum: out[z][y][x] = sum_v Y[v][z][y][x + v] + bias (then the un-normalisation): the kx taps of a
// 1-output-channel conv were computed as K virtual output channels over W + 2*pad columns.
// ------------------------------------------------------------------------------------------
// (only rows [r0, r1) x columns [x0, x1) are produced: what the window of the stem conv reads, rt_exec.hip)
__global__ __launch_bounds__(256) void shiftx_split_kernel(const float* __restrict__ in, uint4* __restrict__ out,
                                                           int ncell, int K, int pad, size_t rows, int W, int Wo,
                                                           unsigned* flag, size_t r0, size_t r1, int x0, int x1) {
    const size_t n = (size_t)ncell * rows * Wo;
    const int nx = x1 - x0;
    const size_t nr = r1 - r0, nw = (size_t)ncell * nr * nx;
    bool big = false;
    for (size_t iw = blockIdx.x * (size_t)blockDim.x + threadIdx.x; iw < nw; iw += (size_t)gridDim.x * blockDim.x) {
        const int x = x0 + (int)(iw % nx);
        const size_t t = iw / nx;
        const size_t row = r0 + t % nr;
        const int jc = (int)(t / nr);
        const size_t i = ((size_t)jc * rows + row) * Wo + x;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * jc + j, sx = x - pad + tap;
            v[j] = (tap < K && (unsigned)sx < (unsigned)W) ? in[row * W + sx] : 0.f;
            big |= !(fabsf(v[j]) <= SPLIT_MAX);
        }
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        uint2 h0, l0, h1, l1;
        split4(a, h0, l0);
        split4(b, h1, l1);
        out[i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[n + i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    if (big && flag) atomicOr(flag, 1u);      // pixel values beyond the f16 range: the caller re-runs in fp32
}

// (only rows [y0, y1) x columns [x0, x1) of the planes [z0, z1) of Hp rows each are produced: the window of the layer,
// rt_exec.hip; a 2-D tensor is one plane of `rows` rows)
__global__ __launch_bounds__(256) void shiftsum_kernel(const float* __restrict__ Y, float* __restrict__ out, int K,
                                                       size_t rows, int W, int Wp, float bias,
                                                       const float* __restrict__ nrm, int norm_out, size_t y0, size_t y1,
                                                       int x0, int x1, const float* __restrict__ res, size_t Hp, int z0, int z1) {
    const int nx = x1 - x0;
    const size_t ny = y1 - y0, n = (size_t)(z1 - z0) * ny * nx;
    float sc = 1.f, sh = 0.f;
    if (nrm && norm_out) { sc = nrm[2]; sh = nrm[3]; }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = x0 + (int)(i % nx);
        const size_t t = i / nx;
        const size_t row = (z0 + t / ny) * Hp + y0 + t % ny;
        float acc = 0.f;
        for (int v = 0; v < K; ++v) acc += Y[((size_t)v * rows + row) * Wp + x + v];
        if (res) acc += res[row * W + x];          // (same-size residual: UDenoiseNet3's x - dec1(h), weights negated)
        out[row * W + x] = (acc + bias) * sc + sh;
    }
}

hipError_t launch_shiftx_split(const float* in, void* out, int K, int pad, size_t rows, int W, int Wo, unsigned* flag,
                               hipStream_t s, size_t r0, size_t r1, int x0, int x1) {
    const int ncell = (K + 7) / 8;
    if (r1 > rows) r1 = rows;
    if (x1 > Wo) x1 = Wo;
    const size_t n = (size_t)ncell * (r1 - r0) * (size_t)(x1 - x0);
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(shiftx_split_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, in, (uint4*)out, ncell, K, pad, rows, W,
                       Wo, flag, r0, r1, x0, x1);
    return hipGetLastError();
}
hipError_t launch_shiftsum(const float* Y, float* out, int K, size_t rows, int W, int Wp, float bias, const float* nrm,
                           int norm_out, hipStream_t s, size_t y0, size_t y1, int x0, int x1, const float* res, int Hp, int z0,
                           int z1) {
    size_t hp = rows;
    if (Hp > 0) hp = (size_t)Hp; else { z0 = 0; z1 = 1; }
    if (y1 > hp) y1 = hp;
    if (x1 > W) x1 = W;
    const size_t n = (size_t)(z1 - z0) * (y1 - y0) * (size_t)(x1 - x0);
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(shiftsum_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, Y, out, K, rows, W, Wp, bias, nrm, norm_out,
                       y0, y1, x0, x1, res, hp, z0, z1);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 1-output-channel last conv of the denoisers on the 2xf16 path (denoising/models.py:125-128, 238-244, 560-562: Conv(32, 1, k)):
// split cells in, ONE fp32 plane out.  On the matrix cores this layer is an M = 16 tile with one useful row per kx tap
// (conv_split_kernel<K x 1, MT = 16>: 26 TFLOP/s, + the shift-sum of its k planes); here it is what it is -- an HBM-bound
// stencil of cin * k^dims multiply-adds per pixel -- on the vector ALUs in full fp32:
//   * tile = TH x 64 output pixels per 256-thread workgroup, two workgroups per CU; wave w owns R = TH / 4 consecutive rows,
//     lane l column l: a lane's 16-byte LDS reads are those of its neighbours shifted by one pixel (conflict-free);
//   * per (input plane kz, 8-channel cell): the (TH + K - 1) x (64 + K - 1) input pixels of the tile are fetched with buffer
//     loads (out-of-image = zeros), hi + lo joined ONCE per pixel (exact: the halves carry 22 bits) and stored as fp32,
//     [half of the cell][row][pixel] x 16 B;
//   * per kx a lane holds its column of R + K - 1 pixels x 8 channels in registers and runs the K ky taps of its R outputs
//     over them with v_pk_fma_f32 on channel pairs; the taps' weights ([kz][cell][kx][ky][8] fp32) arrive as scalar loads.
// A pixel's sum is formed in a fixed order (kz, cell, kx, ky, channel pair) whatever tile or window it falls into: patch
// windows, batched passes and tile windows stay bit-identical.  Bias, the same-size residual (UDenoiseNet3) and the
// un-normalisation of Denoise._denoise (denoise.py:291-294) are applied in the same pass: no k-plane scratch tensor, no
// shift-sum launch.
// ------------------------------------------------------------------------------------------
template <int K, int TH, int ZP>
__global__ __launch_bounds__(256, 2) void conv_cout1_split_kernel(const uint4* __restrict__ in, const float* __restrict__ wt,
                                                                  float* __restrict__ out, const float* __restrict__ res,
                                                                  const float* __restrict__ nrm, int norm_out, float bias,
                                                                  int cells, int KZ, int D, int H, int W, int pad, int wz0,
                                                                  int wz1, int wy0, int wx0, int wy1, int wx1, int tiles_x,
                                                                  int tiles_y) {
    constexpr int TW = 64, R = TH / 4, IR = R + K - 1, ITH = TH + K - 1, ITW = TW + K - 1;
    constexpr int HALF = ITH * ITW;                 // 16-byte slots per half-cell plane of the LDS tile
    constexpr int NJ = (HALF + 255) / 256;          // input pixels a thread stages per (plane, cell)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float4* const lds = reinterpret_cast<float4*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    // the workgroup's ZP output planes z0 .. z0 + ZP - 1 (3-D: every staged input plane serves up to min(ZP, K) of them -- the
    // planes of a k^3 stencil are re-read k times otherwise; 2-D: ZP = 1)
    const int z0 = wz0 + (int)blockIdx.y * ZP;
    const int y0 = wy0 + by * TH, x0 = wx0 + bx * TW;
    const size_t plane_px = (size_t)H * W, cell_px = (size_t)D * plane_px;
    const size_t lo_off = (size_t)cells * cell_px;             // cells from a hi cell to its lo cell
    // the tile's input pixels this thread stages: their offsets within a plane (the same for every plane and cell); -1 outside
    // the image and outside what the launch window's outputs read (a tile may overhang the window: those outputs are not stored)
    int poff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = tid + j * 256;
        const int r = i / ITW, x = i - r * ITW;
        const int gy = y0 - pad + r, gx = x0 - pad + x;
        const bool ok = i < HALF && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && gy < wy1 + pad && gx < wx1 + pad;
        poff[j] = ok ? gy * W + gx : -1;
    }
    // stage s = pi * cells + c: cell c of input plane z0 + pi - pad_z.  Its pixels travel global -> registers (issued one stage
    // AHEAD, in flight under the arithmetic of the current stage) -> joined to fp32 -> LDS.
    const int pad_z = KZ > 1 ? pad : 0;
    const int S = (ZP + KZ - 1) * cells;
    uint4 gh[NJ], gl[NJ];
    auto fetch = [&](int s) {
        const int pi = s / cells, c = s - pi * cells, iz = z0 + pi - pad_z;
        const bool plane_ok = (unsigned)iz < (unsigned)D && iz < wz1 + pad_z;     // (zero padding in z; planes past the window)
        const uint4* src = in + ((size_t)c * D + (plane_ok ? iz : 0)) * plane_px;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            gh[j] = make_uint4(0, 0, 0, 0);
            gl[j] = make_uint4(0, 0, 0, 0);
            if (plane_ok && poff[j] >= 0) {
                gh[j] = src[poff[j]];
                gl[j] = src[(size_t)poff[j] + lo_off];
            }
        }
    };
    f32x2 acc[ZP][R];
#pragma unroll
    for (int o = 0; o < ZP; ++o)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[o][r] = (f32x2){0.f, 0.f};
    fetch(0);
    for (int s = 0; s < S; ++s) {
        __syncthreads();                                       // the previous stage's readers are done
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int i = tid + j * 256;
            if (i < HALF) {
                const f32x2 v0 = join2m(gh[j].x, gl[j].x), v1 = join2m(gh[j].y, gl[j].y);
                const f32x2 v2 = join2m(gh[j].z, gl[j].z), v3 = join2m(gh[j].w, gl[j].w);
                lds[i] = make_float4(v0[0], v0[1], v1[0], v1[1]);
                lds[HALF + i] = make_float4(v2[0], v2[1], v3[0], v3[1]);
            }
        }
        __syncthreads();
        if (s + 1 < S) fetch(s + 1);
        const int pi = s / cells, c = s - pi * cells;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            f32x2 px[IR][4];
#pragma unroll
            for (int i = 0; i < IR; ++i) {
                const int slot = (wave * R + i) * ITW + lane + kx;
                const float4 a = lds[slot], b = lds[HALF + slot];
                px[i][0] = (f32x2){a.x, a.y}; px[i][1] = (f32x2){a.z, a.w};
                px[i][2] = (f32x2){b.x, b.y}; px[i][3] = (f32x2){b.z, b.w};
            }
#pragma unroll
            for (int o = 0; o < ZP; ++o) {
                const int kz = pi - o;                             // output plane z0 + o takes this input plane as its tap kz
                if (kz < 0 || kz >= KZ) continue;                  // (wave-uniform)
                const float* wc = wt + ((size_t)kz * cells + c) * (K * K * 8);
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const float* w8 = wc + (kx * K + ky) * 8;      // wave-uniform: scalar loads
                    const f32x2 w0 = {w8[0], w8[1]}, w1 = {w8[2], w8[3]}, w2 = {w8[4], w8[5]}, w3 = {w8[6], w8[7]};
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[o][r] = __builtin_elementwise_fma(w0, px[r + ky][0], acc[o][r]);
                        acc[o][r] = __builtin_elementwise_fma(w1, px[r + ky][1], acc[o][r]);
                        acc[o][r] = __builtin_elementwise_fma(w2, px[r + ky][2], acc[o][r]);
                        acc[o][r] = __builtin_elementwise_fma(w3, px[r + ky][3], acc[o][r]);
                    }
                }
            }
        }
    }
    float sc = 1.f, sh = 0.f;
    if (nrm && norm_out) { sc = nrm[2]; sh = nrm[3]; }
    const int ox = x0 + lane;
#pragma unroll
    for (int o = 0; o < ZP; ++o) {
        const int z = z0 + o;
        if (z >= wz1) continue;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int oy = y0 + wave * R + r;
            if (oy < wy1 && ox < wx1) {
                const size_t idx = ((size_t)z * H + oy) * W + ox;
                float v = acc[o][r][0] + acc[o][r][1];
                if (res) v += res[idx];
                out[idx] = (v + bias) * sc + sh;
            }
        }
    }
}

// in: split cells [2][cells][D][H][W]; wt: [KZ][cells][K(kx)][K(ky)][8] fp32; out / res: fp32 [D][H][W]; the launch covers the
// planes [z0, z1) x rows [y0, y1) x columns [x0, x1) of the output (pad = K / 2: same-size convolution)
template <int K, int TH, int ZP>
static hipError_t launch_cout1_cfg(const void* in, const float* wt, float* out, const float* res, const float* nrm, int norm_out,
                                   float bias, int cells, int KZ, int D, int H, int W, int z0, int z1, int y0, int y1, int x0,
                                   int x1, hipStream_t s) {
    constexpr int LDSB = 2 * (TH + K - 1) * (64 + K - 1) * 16;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_cout1_split_kernel<K, TH, ZP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (e != hipSuccess) return e;
        attr = true;
    }
    const int tiles_x = (x1 - x0 + 63) / 64, tiles_y = (y1 - y0 + TH - 1) / TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)((z1 - z0 + ZP - 1) / ZP), 1);
    hipLaunchKernelGGL((conv_cout1_split_kernel<K, TH, ZP>), grid, dim3(256), LDSB, s, (const uint4*)in, wt, out, res, nrm, norm_out,
                       bias, cells, KZ, D, H, W, K / 2, z0, z1, y0, x0, y1, x1, tiles_x, tiles_y);
    return hipGetLastError();
}

hipError_t launch_conv_cout1_split(const void* in, const float* wt, float* out, const float* res, const float* nrm, int norm_out,
                                   float bias, int cells, int K, int KZ, int D, int H, int W, int z0, int z1, int y0, int y1,
                                   int x0, int x1, hipStream_t s) {
    if (z1 <= z0 || y1 <= y0 || x1 <= x0) return hipSuccess;
    if (KZ != 1 && KZ != K) return hipErrorInvalidValue;
    // a pixel's sum is formed in the same order whatever ZP: the choice is free (2-D: one plane; 3-D: two output planes per
    // workgroup share four staged input planes instead of six)
    if (K == 5 && KZ == 1) return launch_cout1_cfg<5, 32, 1>(in, wt, out, res, nrm, norm_out, bias, cells, KZ, D, H, W, z0, z1, y0, y1, x0, x1, s);
    if (K == 3 && KZ == 1) return launch_cout1_cfg<3, 32, 1>(in, wt, out, res, nrm, norm_out, bias, cells, KZ, D, H, W, z0, z1, y0, y1, x0, x1, s);
    if (K == 3) {
        // (ZP = 1 / 4 were measured in round 5, profiles/r05_last_conv_ab.txt: 2 it is)
        return launch_cout1_cfg<3, 32, 2>(in, wt, out, res, nrm, norm_out, bias, cells, KZ, D, H, W, z0, z1, y0, y1, x0, x1, s);
    }
    if (K == 5) return launch_cout1_cfg<5, 32, 1>(in, wt, out, res, nrm, norm_out, bias, cells, KZ, D, H, W, z0, z1, y0, y1, x0, x1, s);
    return hipErrorInvalidValue;
}

// space-to-depth of a 1-channel image / volume into one split cell per low-resolution pixel: channel
// j = (2*qz + qy)*2 + qx of cell (z, y, x) is in[2z + qz][2y + qy][2x + qx] (2-D: qz = 0, channels 4..7 zero).  The
// 1-channel skip source of a U-Net's dec1.0 joins the per-parity form of that layer as these extra input channels
// (rt_load.hip prepare_split_phases).
__global__ __launch_bounds__(256) void s2d_split_kernel(const float* __restrict__ in, uint4* __restrict__ out, int d, int h,
                                                        int w, int H, int W, int dims, unsigned* flag) {
    const size_t n = (size_t)d * h * w;
    bool big = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        const size_t t = i / w;
        const size_t y = t % h, z = t / h;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int nz = dims == 3 ? 2 : 1;
        for (int qz = 0; qz < nz; ++qz) {
            const float* p = in + (((dims == 3 ? 2 * z + qz : 0) * (size_t)H + 2 * y) * W + 2 * x);
            const float2 r0 = *reinterpret_cast<const float2*>(p);
            const float2 r1 = *reinterpret_cast<const float2*>(p + W);
            v[qz * 4 + 0] = r0.x; v[qz * 4 + 1] = r0.y; v[qz * 4 + 2] = r1.x; v[qz * 4 + 3] = r1.y;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) big |= !(fabsf(v[j]) <= SPLIT_MAX);
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        uint2 h0, l0, h1, l1;
        split4(a, h0, l0);
        split4(b, h1, l1);
        out[i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[n + i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    if (big && flag) atomicOr(flag, 1u);
}

hipError_t launch_s2d_split(const float* in, void* out, int d, int h, int w, int H, int W, int dims, unsigned* flag,
                            hipStream_t s) {
    const size_t n = (size_t)d * h * w;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(s2d_split_kernel, dim3(blocks), dim3(256), 0, s, in, (uint4*)out, d, h, w, H, W, dims, flag);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// out[c][r] = in[r][c] through a padded 64x64 LDS tile (both sides coalesced)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                        int Cc) {
    __shared__ float t[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < R && c < Cc) ? in[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int c = c0 + k, r = r0 + tx;
        if (c < Cc && r < R) out[(size_t)c * R + r] = t[tx][k];
    }
}

hipError_t launch_transpose(const float* in, float* out, int R, int Cc, hipStream_t s) {
    if (R <= 0 || Cc <= 0) return hipSuccess;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 63) / 64, (R + 63) / 64), dim3(256), 0, s, in, out, R, Cc);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// PatchDataset.__getitem__ (denoising/datasets.py:426-468) fused with the global normalisation
// (denoise.py:355): tile[d][d][d] = ((inside ? tomo : 0) - mu)/std, with g = {mu, std} on device.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void extract_tile3d_kernel(const float* __restrict__ tomo, int D, int H, int W,
                                                             int i0, int j0, int k0, int d, const float* __restrict__ g,
                                                             float* __restrict__ tile) {
    const size_t n = (size_t)d * d * d;
    const float mu = g[0], sd = g[1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % d);
        const size_t t = i / d;
        const int y = (int)(t % d);
        const int z = (int)(t / d);
        const int gz = i0 + z, gy = j0 + y, gx = k0 + x;
        float v = 0.f;
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = tomo[((size_t)gz * H + gy) * W + gx];
        tile[i] = (v - mu) / sd;
    }
}

hipError_t launch_extract_tile3d(const float* tomo, int D, int H, int W, int i0, int j0, int k0, int d,
                                 const float* d_g, float* tile, hipStream_t s) {
    const size_t n = (size_t)d * d * d;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(extract_tile3d_kernel, dim3(blocks), dim3(256), 0, s, tomo, D, H, W, i0, j0, k0, d, d_g, tile);
    return hipGetLastError();
}

}  // namespace tpz
