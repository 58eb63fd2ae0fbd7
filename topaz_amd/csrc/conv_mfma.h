// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Replaces the torch.nn.Conv2d / Conv3d calls the reference issues on its hot path
// (topaz/model/features/resnet.py:129-133,294-302, topaz/model/features/basic.py:47-63,
//  topaz/denoising/models.py:79-128,457-512): dilated "valid" convs of the filled scoring
// nets and "same" convs of the U-Nets, with the bias / activation / residual / eval-BN /
// 1x1-head epilogue fused (resnet.py:101-105,185-202, classifier.py:64-66) and the
// nearest-upsample + concat of the U-Net decoders folded into the loader (models.py:140-171).
//
// GEMM view: M = output channels, N = output pixels, K = Cin * taps.
//   one MFMA: A[16 co][4 k] * B[4 k][16 px] -> C[16 co][16 px], exact f32 (fmaf chain).
//   k-group of 4 = four consecutive input channels at one tap            (generic)
//                = four consecutive kx taps of the single input channel  (CIN1 stems)
// Workgroup = 256 threads = 4 waves (one per SIMD); tile = MT output channels x (TD x TH x TW)
//   pixels.  Rows (and planes) of the tile are strided by the dilation D ("polyphase" in y/z),
//   so the LDS halo in y/z is K-1 rows instead of (K-1)*D; columns are contiguous with a
//   (K-1)*D halo so every global row segment is a coalesced read.
// Pipeline: the K loop is cut into stages = (a chunk of 4*KG input channels) x (RPS tap rows).
//   Both operands of stage s+1 are fetched by the LDS-DMA path (global_load_lds, no VGPRs)
//   into the second LDS buffer while the MFMAs of stage s run; one barrier per stage.
//   Weights are pre-packed on the host in A-fragment lane order, so the A read is
//   lds[step][mf][lane], the B read is lds[ch][z][y][x]; all ds_read offsets are immediates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "split_fmt.h"

namespace tpz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* in;          // [Cin1][D1][H1][W1] (strided)
    const float* in2;         // optional 2nd source: channels [Cin1, Cin) come from here (fused concat)
    const float* wpk;         // packed weights (rt_load.hip pack_weights)
    const float* bias;        // [Cout] or nullptr
    float* out;               // [Cout][Dout][Hout][Wout] (nullptr when the head is fused)
    const float* res;         // residual [Cout][Dres][Hres][Wres] or nullptr
    const float* post_scale;  // [Cout] affine applied after the residual add (eval BN), or nullptr
    const float* post_shift;
    const float* head_w;      // fused 1x1 head: [Cout] weights, or nullptr
    float* head_out;          // [Dout][Hout][Wout]
    const float* zeros;       // >= 16 bytes of zeros in global memory (source of padded / OOB elements)
    const float* nrm;         // device float[4] {in_scale, in_shift, out_scale, out_shift} or nullptr
    unsigned* flag;           // EPI_SPLIT: set to 1 when a stored value leaves the f16 range
    float head_b;
    int norm_out;             // y' = y*out_scale+out_shift applied last
    int Cin, Cin1;            // Cin1 = channels taken from `in` (== Cin when no concat)
    int Din, Hin, Win;        // logical input geometry (== geometry of in2; `in` is nearest-upsampled to it)
    int D1, H1, W1;           // geometry of `in`
    long long cs1, ps1; int pitch1;   // channel / plane / row strides (floats) of `in`
    long long cs2, ps2; int pitch2;   // ... of `in2`
    int cog_inner;            // co-groups looped inside the kernel (fused head), else 1
    int Cout, Dout, Hout, Wout;
    int pad;                  // zero padding on every side (direct kernels); the MFMA kernel uses the per-axis values
    int pad_x, pad_y, pad_z;
    // Output lattice: element (oz, oy, ox) of the launch is stored at (oz*os + ooz, oy*os + ooy, ox*os + oox) of
    // a [Cout][Dfull][Hfull][Wfull] tensor (os = 1, offsets 0, full = out dims for an ordinary convolution;
    // os = 2 for the phase launches of a conv over an exactly 2x nearest-upsampled source, see rt_load.hip prepare_phases).
    int os, ooz, ooy, oox;
    int Dfull, Hfull, Wfull;
    int Dres, Hres, Wres, res_crop;
    int n_chunks;             // channel chunks of NCH channels
    float slope;              // activation: v > 0 ? v : v*slope   (1.0 = identity, 0.0 = ReLU)
    int tiles_x, tiles_y, tiles_z;
    // launch window (2-D): the outputs [wy0, wy1) x [wx0, wx1) of the launch lattice are computed and stored, the tiles start
    // at (wy0, wx0).  Whole tensor: 0, 0, Hout, Wout (conv_window_default; wy1 == 0 means "not set").  wx0 % 4 == 0 keeps the
    // 16-byte granules of the x4 loader aligned.
    int wy0, wx0, wy1, wx1;
    // ... and, 3-D, the planes [wz0, wz1) of the launch lattice (a tile of a tiled tomogram keeps its centre in z as well);
    // wz1 == 0: not set (launch_mfma / launch_conv_direct default it to [0, Dout))
    int wz0, wz1;
    int xcd_swizzle;          // 1: remap workgroup ids so that each XCD owns a contiguous run of tiles
    int stagger_first;        // workgroups with a linear id below this belong to the first generation
    int stagger_sleeps;       // s_sleep(127) repeats for the odd wave slot of the first generation (0 = off)
};

// PyTorch 'nearest' source index: min(floor(dst * (float)in/out), in-1)  
__device__ __forceinline__ int nearest_src(int dst, int in_sz, int out_sz) {
    if (in_sz == out_sz) return dst;
    float scale = (float)in_sz / (float)out_sz;
    int s = (int)floorf((float)dst * scale);
    return s < in_sz - 1 ? s : in_sz - 1;
}

template <int K_, int D_, int MT_, int TD_, int TH_, int TW_, int KG_, int RPS_, bool CIN1_, int DIMS_>
struct ConvCfg {
    static constexpr int K = K_, D = D_, MT = MT_, TD = TD_, TH = TH_, TW = TW_, KG = KG_, RPS = RPS_, DIMS = DIMS_;
    static constexpr bool CIN1 = CIN1_;
    static constexpr int KZ = (DIMS == 3) ? K : 1;
    static constexpr int MW = MT / 16;
    static constexpr int ROWS = TD * TH;            // tile rows (z-major)
    static constexpr int RPW = ROWS / 4;            // rows per wave
    static constexpr int NFC = TW / 16;             // N fragments per row
    static constexpr int NW = RPW * NFC;
    static constexpr int KP = CIN1 ? ((K + 3) / 4 * 4) : K;   // kx taps padded to a k-group
    static constexpr int KXG = CIN1 ? KP / 4 : K;             // MFMA k-steps per tap row
    static constexpr int ITD = TD + KZ - 1;
    static constexpr int ITH = TH + K - 1;
    // +3: the tile's x origin is moved left to a multiple of 4 pixels (16-byte DMA granules, see kernel)
    static constexpr int ITW = (TW + (KP - 1) * D + 3 + 3) / 4 * 4;
    static constexpr int RS = ITW;
    static constexpr int PS = ITH * RS;             // plane stride
    static constexpr int TILE_ELEMS = ITD * PS;
    // channel stride == 16 (mod 32): the two 16-lane halves of a ds_read_b32 group hit disjoint banks
    static constexpr int CS = CIN1 ? TILE_ELEMS : (((TILE_ELEMS - 16 + 31) / 32) * 32 + 16);
    static constexpr int NCH = CIN1 ? 1 : 4 * KG;
    static constexpr int IN_BUF = ((NCH * CS + 1023) / 1024) * 1024;  // floats per input buffer (DMA granule 256 x 16 B)
    static constexpr int NROWS = KZ * K;            // tap rows (kz, ky)
    static constexpr int SPG = NROWS / RPS;         // stages per channel chunk
    static constexpr int STEPS = (CIN1 ? 1 : KG) * RPS * KXG;         // MFMA k-steps per stage
    static constexpr int W_STAGE = STEPS * MW * 64; // floats per weight stage
    static constexpr int W_CHUNK = SPG * W_STAGE;   // floats per (co-group, chunk)
    // LDS: 2 input buffers, 2 weight stages, and the per-element DMA source-offset table (IN_BUF words)
    static constexpr int LDS_BYTES = (3 * IN_BUF + 2 * W_STAGE) * 4;
    // a wave's RPW tile rows either sit inside one z-plane, or cover whole z-planes
    static constexpr bool IN_PLANE = (TH % RPW == 0);
    static_assert(IN_PLANE || (RPW % TH == 0), "wave rows must align with z-planes");
    static constexpr int row_off(int row, int kz, int ky) {
        return IN_PLANE ? kz * PS + (row + ky) * RS : (row / TH + kz) * PS + (row % TH + ky) * RS;
    }
    static_assert(NROWS % RPS == 0, "stages must tile the tap rows");
    static_assert(RPS == 1 || RPS % K == 0, "a stage is one tap row or whole kz planes");
    static_assert(SPG == 1 || KG == 1 || CIN1, "row-split stages need KG == 1");
    static_assert(ROWS % 4 == 0, "tile rows must split over 4 waves");
    static_assert(TW % 16 == 0 && MT % 16 == 0, "MFMA 16x16 fragments");
    static_assert(IN_BUF * 4 < 65536 && W_STAGE * 4 < 65536, "ds_read immediate offsets are 16 bit");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS per workgroup");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA (global_load_lds) in the SGPR-base + 32-bit VGPR-offset form.  Written as inline asm so that the
// address stays one VGPR per element (hipcc otherwise materialises -- and spills -- 64-bit pointers, and
// waits vmcnt(0) on every reload, serialising the DMA).  Data lands at lds_addr + lane*size.  M0 is
// saved/restored inside the statement (cdna_hip_programming.md 5.7).  hipcc does not count these loads:
// the pipeline waits with an explicit `s_waitcnt vmcnt(0)` before each stage barrier.
__device__ __forceinline__ void glds_b32(unsigned voff, const void* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ void glds_b128(unsigned voff, const void* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}

// ABL: timing-ablation switches used by tools/conv_ablate.hip only (production kernels use ABL = 0):
//   2 skip the per-stage DMA issue   4 skip the per-stage barrier
//   8 fragment loads only for the first step of a stage (operands reused)   16 static wave-slot priority
// EPI selects the epilogue the kernel is compiled for (one lean, branch-free code path each):
//   0 bias + activation            1 + residual add (ResidA skip)
//   2 + residual + eval-BN affine  3 bias + activation + fused 1x1 head
//   4 bias + activation, output stored as split f16 cells for the 2xf16 path (split_fmt.h, conv_split.h)
enum { EPI_PLAIN = 0, EPI_RES = 1, EPI_RES_POST = 2, EPI_HEAD = 3, EPI_SPLIT = 4 };

template <class C, int EPI = 0, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int K = C::K, D = C::D, MW = C::MW, NW = C::NW, NFC = C::NFC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_in = lds;                       // 2 x IN_BUF
    float* lds_w = lds + 2 * C::IN_BUF;        // 2 x W_STAGE
    unsigned* lds_tab = reinterpret_cast<unsigned*>(lds + 2 * C::IN_BUF + 2 * C::W_STAGE);   // IN_BUF offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    // (A static s_setprio split between the two co-resident workgroups was measured: neutral on the
    //  MT=128 tiles, -10 % on MT=64 -- tools/conv_ablate.hip bit 16 re-enables it for experiments.)
    if constexpr ((ABL & 16) != 0) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID.WAVE_ID
        if (hw_id & 1u) __builtin_amdgcn_s_setprio(2);
    }
    // Phase stagger: the two workgroups sharing a CU start together and, running identical code, stay in
    // lockstep -- both in their prologue / epilogue (no MFMA) at the same time, generation after generation.
    // The workgroups of the FIRST generation that sit in the odd hardware wave slot wait half a workgroup
    // lifetime once; from then on one workgroup's fixed costs overlap the other's MFMA phase.
    if (a.stagger_sleeps > 0 && (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) < a.stagger_first) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (hw_id & 1u)
            for (int i = 0; i < a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);   // 127 * 64 cycles each
    }

    // ---- tile coordinates. y (and z) tiles are polyphase: row i of the tile is y0 + i*D.
    // XCD-aware order: consecutive workgroup ids are dispatched round-robin over the 8 XCDs (each with its
    // own L2), so the (x, y) tile is taken from a remapped id that gives every XCD a contiguous run of
    // tiles -- x-neighbours, which share 24 of 56 halo columns, then hit the same L2.  Bijective for any
    // grid size (cdna_hip_programming.md 5.5 T1).  Speed only; no correctness dependence.
    int bx = blockIdx.x, byz = blockIdx.y;
    if (a.xcd_swizzle) {
        const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
        bx = (int)(wgid % gridDim.x);
        byz = (int)(wgid / gridDim.x);
    }
    const int ty = byz % a.tiles_y;
    const int tz = byz / a.tiles_y;
    const int y0 = a.wy0 + (ty / D) * (C::TH * D) + (ty % D);
    const int z0 = (C::DIMS == 3) ? a.wz0 + (tz / D) * (C::TD * D) + (tz % D) : 0;
    const int x0 = a.wx0 + bx * C::TW;
    // The LDS tile starts PADA = roundup4(pad) pixels left of x0, so every 4-float LDS granule maps to a
    // 16-byte aligned global run when the row pitch is a multiple of 4 floats; B reads shift by PADA - pad.
    const int pada = (a.pad_x + 3) & ~3;
    const int xshift = pada - a.pad_x;
    const int ybase = y0 - a.pad_y, xbase = x0 - pada, zbase = (C::DIMS == 3) ? z0 - a.pad_z : 0;

    // per-lane LDS read offsets (floats)
    const int b_lane = xshift + (C::CIN1 ? (l4 * D + l15) : (l4 * C::CS + l15)) +
                       (C::IN_PLANE ? ((wave * C::RPW) / C::TH) * C::PS + ((wave * C::RPW) % C::TH) * C::RS
                                    : wave * (C::RPW / C::TH) * C::PS);
    const bool ups = (a.H1 != a.Hin) || (a.W1 != a.Win) || (a.D1 != a.Din);
    float out_scale = 1.f, out_shift = 0.f;
    if (a.nrm && a.norm_out) { out_scale = a.nrm[2]; out_shift = a.nrm[3]; }

    // ---- LDS-DMA issue helpers ------------------------------------------------------------------
    // Input chunk -> lds_in buffer: element e = i*256 + tid of the padded [NCH][CS] LDS image comes from
    // (channel c, z, y, x).  The per-thread source offsets (relative to the chunk's first channel) do not
    // depend on the chunk, so they are computed ONCE per source tensor and kept in registers: the
    // per-stage issue is one 64-bit add + one global_load_lds per element.  off < 0: padding or outside
    // the image -> the element is fetched from the global zero word.
    constexpr int NI = C::IN_BUF / 256;    // 4-byte DMA pieces per thread and chunk (fallback mode)
    constexpr int NI4 = C::IN_BUF / 1024;  // 16-byte DMA pieces per thread and chunk
    constexpr unsigned OOB = 0xffffffffu;  // table entry of an element / granule outside the image (zero-filled)
    // Source byte offsets (relative to the chunk's first channel) of the LDS image.  They do not depend on
    // the chunk, so they are computed once per source tensor into an LDS table (keeping them in VGPRs next
    // to 128 accumulators made hipcc spill); each stage re-reads its entries.
    //   x4 mode: one entry per 4-float granule.  Needs rows that start 16-byte aligned and a width that is
    //            a multiple of 4 (then a granule is entirely inside or outside the image) and no upsampling.
    //   x1 mode: one entry per float (any geometry; nearest-upsampled sources).
    bool x4mode = false;
    auto compute_offsets = [&](bool second) {
        const float* p = second ? a.in2 : a.in;
        const long long cs = second ? a.cs2 : a.cs1, ps = second ? a.ps2 : a.ps1;
        const int pitch = second ? a.pitch2 : a.pitch1;
        x4mode = ((a.Win | pitch) % 4 == 0) && (cs % 4 == 0) && (ps % 4 == 0) && (((size_t)p & 15) == 0) &&
                 (second || !ups);
        const int n = x4mode ? NI4 : NI;
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            const int g = i * 256 + tid;
            const int e = x4mode ? 4 * g : g;
            const int c = e / C::CS;
            const int rem = e - c * C::CS;
            const int zz = rem / C::PS;
            const int rem2 = rem - zz * C::PS;
            const int r = rem2 / C::RS;
            const int x = rem2 - r * C::RS;
            const int gy = ybase + r * D, gx = xbase + x;
            const int gz = (C::DIMS == 3) ? zbase + zz * D : 0;
            unsigned off = 0;              // LDS padding is never read: it fetches the chunk's first words
            if (rem < C::TILE_ELEMS && c < C::NCH) {
                if ((unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win && (unsigned)gz < (unsigned)a.Din) {
                    long long o;
                    if (!second) {
                        int sy = gy, sx = gx, sz = gz;
                        if (ups) {
                            sy = nearest_src(gy, a.H1, a.Hin);
                            sx = nearest_src(gx, a.W1, a.Win);
                            if (C::DIMS == 3) sz = nearest_src(gz, a.D1, a.Din);
                        }
                        o = (long long)c * a.cs1 + (long long)sz * a.ps1 + (long long)sy * a.pitch1 + sx;
                    } else {
                        o = (long long)c * a.cs2 + (long long)gz * a.ps2 + (long long)gy * a.pitch2 + gx;
                    }
                    off = (unsigned)(o * 4);
                } else {
                    off = OOB;
                }
            }
            lds_tab[g] = off;              // read back by the same thread only
        }
    };
    // chunks [0, chunks1) read `in`, the rest read `in2` (the host guarantees Cin1 % NCH == 0 with a concat)
    const int chunks1 = (a.in2 != nullptr) ? a.Cin1 / C::NCH : a.n_chunks;
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;     // LDS byte address of the dynamic region
    auto issue_input = [&](int ch, int buf) {
        const bool second = ch >= chunks1;
        const void* base = uniform_ptr(second ? a.in2 + (long long)(ch - chunks1) * C::NCH * a.cs2
                                              : a.in + (long long)ch * C::NCH * a.cs1);
        const void* zbase = uniform_ptr(a.zeros);
        const int c_left = a.Cin - ch * C::NCH;      // channels of this chunk that exist (>= NCH except in the last)
        if (x4mode) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * C::IN_BUF + wave * 256) * 4u);
            unsigned off[NI4];
            bool any_oob = false;
#pragma unroll
            for (int i = 0; i < NI4; ++i) {
                off[i] = lds_tab[i * 256 + tid];
                if (c_left < C::NCH && (4 * (i * 256 + tid)) / C::CS >= c_left) off[i] = OOB;
                any_oob |= (off[i] == OOB);
            }
            if (!__any(any_oob)) {
#pragma unroll
                for (int i = 0; i < NI4; ++i) glds_b128(off[i], base, dst + i * 4096);
            } else {
#pragma unroll
                for (int i = 0; i < NI4; ++i) {
                    if (off[i] != OOB) glds_b128(off[i], base, dst + i * 4096);
                    else glds_b128(0u, zbase, dst + i * 4096);
                }
            }
        } else {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * C::IN_BUF + wave * 64) * 4u);
#pragma unroll 1
            for (int i0 = 0; i0 < NI; i0 += 8) {        // batches of 8 keep the register footprint small
                unsigned off[8];
                bool any_oob = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k;
                    off[k] = i < NI ? lds_tab[i * 256 + tid] : 0u;
                    if (c_left < C::NCH && (i * 256 + tid) / C::CS >= c_left) off[k] = OOB;
                    any_oob |= (off[k] == OOB);
                }
                const bool slow = __any(any_oob);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k;
                    if (i < NI) {
                        if (!slow || off[k] != OOB) glds_b32(off[k], base, dst + i * 1024);
                        else glds_b32(0u, zbase, dst + i * 1024);
                    }
                }
            }
        }
    };
    // weight stage `st` (global stage index within the co-group) -> lds_w buffer `buf`
    auto issue_weights = [&](const float* wcog, int st, int buf) {
        const void* base = uniform_ptr(wcog + (size_t)st * C::W_STAGE);
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(2 * C::IN_BUF + buf * C::W_STAGE + wave * 256) * 4u);
        constexpr int N4 = C::W_STAGE / 4;
#pragma unroll
        for (int i = 0; i < (N4 + 255) / 256; ++i) {
            if (i * 256 + tid < N4) glds_b128((unsigned)(i * 256 + tid) * 16u, base, dst + i * 4096);
        }
    };

    float hsum[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) hsum[n] = 0.f;

    const int n_stages = a.n_chunks * C::SPG;

    for (int cg = 0; cg < a.cog_inner; ++cg) {
        const int cog = blockIdx.z * a.cog_inner + cg;
        const float* wcog = a.wpk + (size_t)cog * a.n_chunks * C::W_CHUNK;
        f32x4 acc[MW][NW];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

        __syncthreads();                       // previous co-group done with the LDS buffers
        compute_offsets(chunks1 == 0);
        issue_input(0, 0);
        issue_weights(wcog, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int s = 0; s < n_stages; ++s) {
            const int ch = s / C::SPG;         // SPG is a compile-time constant
            const int j = s - ch * C::SPG;
            // ---- prefetch stage s+1 by DMA while this stage computes
            if constexpr (!(ABL & 2)) {
                if (s + 1 < n_stages) issue_weights(wcog, s + 1, (s + 1) & 1);
                if (j == 0 && ch + 1 < a.n_chunks) {
                    if (ch + 1 == chunks1) compute_offsets(true);    // switching to the concatenated source
                    issue_input(ch + 1, (ch + 1) & 1);
                }
            }

            // ---- MFMAs of stage s
            const int row0 = j * C::RPS;       // first tap row of the stage
            const float* bl = lds_in + (ch & 1) * C::IN_BUF + b_lane +
                              (C::RPS == 1 ? (row0 / K) * C::PS + (row0 % K) * C::RS : (row0 / K) * C::PS);
            const float* al = lds_w + (s & 1) * C::W_STAGE + lane;
            // fragment loads of step t+1 are issued ahead of the MFMAs of step t (register double buffer)
            float av[2][MW], bv[2][NW];
            auto load_frags = [&](int step, float (&a_)[MW], float (&b_)[NW]) {
                const int kx = step % C::KXG;
                const int r = (step / C::KXG) % C::RPS;
                const int kg = step / (C::KXG * C::RPS);
                const int rz = r / K, ry = r % K;                   // stage-relative (kz, ky)
#pragma unroll
                for (int m = 0; m < MW; ++m) a_[m] = al[(step * MW + m) * 64];
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    const int row = n / NFC, cc = n % NFC;
                    const int off = C::row_off(row, rz, ry) + cc * 16 +
                                    (C::CIN1 ? kx * 4 * D : kg * 4 * C::CS + kx * D);
                    b_[n] = bl[off];
                }
            };
            load_frags(0, av[0], bv[0]);
#pragma unroll
            for (int step = 0; step < C::STEPS; ++step) {
                if constexpr (!(ABL & 8)) {
                    if (step + 1 < C::STEPS) load_frags(step + 1, av[(step + 1) & 1], bv[(step + 1) & 1]);
                } else {
#pragma unroll
                    for (int m = 0; m < MW; ++m) av[(step + 1) & 1][m] = av[step & 1][m];
#pragma unroll
                    for (int n = 0; n < NW; ++n) bv[(step + 1) & 1][n] = bv[step & 1][n];
                }
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[step & 1][m], bv[step & 1][n], acc[m][n],
                                                                         0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of stage s+1 has landed
            if constexpr (!(ABL & 4)) __syncthreads();         // ... everyone's has; stage s buffers are free
        }

        // ---- epilogue: bias, residual, eval-BN affine, activation, (fused 1x1 head), store.
        // One predicate per 16-pixel fragment (not per element) and clamped channel indices keep the
        // scattered loads of a fragment free of branches, so they are issued back to back.
        const size_t plane_out = (size_t)a.Hfull * a.Wfull;
        const size_t vol_out = plane_out * a.Dfull;
        const size_t plane_res = (size_t)a.Hres * a.Wres;
        const size_t vol_res = plane_res * a.Dres;
        const int co0 = cog * C::MT + l4 * 4;
        const bool has_bias = a.bias != nullptr;
        bool big = false;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int trow = wave * C::RPW + n / NFC;             // tile row, z-major
            const int ti_z = trow / C::TH, ti_y = trow % C::TH;
            const int oy = y0 + ti_y * D, oz = (C::DIMS == 3) ? z0 + ti_z * D : 0;
            const int ox = x0 + (n % NFC) * 16 + l15;
            if ((oy < a.wy1) && (ox < a.wx1) && (oz < a.wz1)) {
                // position in the full output tensor (identity unless this is a phase launch)
                const int fz = oz * a.os + a.ooz, fy = oy * a.os + a.ooy, fx = ox * a.os + a.oox;
                const size_t pix_res = (size_t)(C::DIMS == 3 ? fz + a.res_crop : 0) * plane_res +
                                       (size_t)(fy + a.res_crop) * a.Wres + (fx + a.res_crop);
                const size_t pix_out = (size_t)fz * plane_out + (size_t)fy * a.Wfull + fx;
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + m * 16 + r;
                        const int cc = co < a.Cout ? co : a.Cout - 1;          // clamped: loads stay in range
                        v[r] = acc[m][n][r] + (has_bias ? a.bias[cc] : 0.f);
                        if constexpr (EPI == EPI_RES || EPI == EPI_RES_POST) v[r] += a.res[(size_t)cc * vol_res + pix_res];
                        if constexpr (EPI == EPI_RES_POST) v[r] = v[r] * a.post_scale[cc] + a.post_shift[cc];
                        v[r] = v[r] > 0.f ? v[r] : v[r] * a.slope;
                        if constexpr (EPI == EPI_HEAD) v[r] *= (co < a.Cout ? a.head_w[cc] : 0.f);
                    }
                    if constexpr (EPI == EPI_HEAD) {
                        hsum[n] += (v[0] + v[1]) + (v[2] + v[3]);
                    } else if constexpr (EPI == EPI_SPLIT) {
                        // the lane's 4 consecutive channels are half of a 16-byte cell (hi plane, then lo plane)
                        const int c4 = co0 + m * 16;
                        const int cell = c4 >> 3, half = (c4 >> 2) & 1;
                        const int cells_out = (a.Cout + 7) >> 3;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (c4 + r >= a.Cout) v[r] = 0.f;
                            big |= !(fabsf(v[r]) <= SPLIT_MAX);      // also true for NaN
                        }
                        if (cell < cells_out) {
                            uint2 hi, lo;
                            split4(v, hi, lo);
                            uint2* op = reinterpret_cast<uint2*>(reinterpret_cast<uint4*>(a.out) + (size_t)cell * vol_out + pix_out) + half;
                            op[0] = hi;
                            op[(size_t)cells_out * vol_out * 2] = lo;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = co0 + m * 16 + r;
                            float w_ = v[r];
                            if (a.norm_out) w_ = w_ * out_scale + out_shift;
                            if (co < a.Cout) a.out[(size_t)co * vol_out + pix_out] = w_;
                        }
                    }
                }
            }
        }
        if constexpr (EPI == EPI_SPLIT) {
            if (__any(big) && lane == 0) atomicOr(a.flag, 1u);
        }
    }  // co-group loop

    if constexpr (EPI == EPI_HEAD) {
        // fused 1x1 head: reduce over the four 16-lane groups (they hold different co of the same pixel)
        const size_t plane_o = (size_t)a.Hfull * a.Wfull;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int trow = wave * C::RPW + n / NFC;
            const int ti_z = trow / C::TH, ti_y = trow % C::TH;
            const int oy = y0 + ti_y * D, oz = (C::DIMS == 3) ? z0 + ti_z * D : 0;
            const int ox = x0 + (n % NFC) * 16 + l15;
            float h = hsum[n];
            h += __shfl_xor(h, 16, 64);
            h += __shfl_xor(h, 32, 64);
            if (l4 == 0 && oy < a.wy1 && ox < a.wx1 && oz < a.wz1) {
                h += a.head_b;
                if (a.norm_out) h = h * out_scale + out_shift;
                a.head_out[(size_t)(oz * a.os + a.ooz) * plane_o + (size_t)(oy * a.os + a.ooy) * a.Wfull + (ox * a.os + a.oox)] = h;
            }
        }
    }
}

}  // namespace tpz
