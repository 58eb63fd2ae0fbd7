// Weights-resident 2xf16 convolution for the narrow ResidA layers of the 32-unit detectors: 3x3, 32 -> 32 channels, dilation
// 1 / 2 / 4 (topaz/model/features/resnet.py:108-204 filled; nine of the 17 layers of the shipped resnet16_u32, three of the 9 of
// resnet8_u32).
//
// On conv_split_kernel these layers are bound by everything that is not an MFMA: a tile's K loop is 9 steps of
// 24 MFMAs per wave, each step paying its weight DMA, its share of the input DMA, a plan entry and a workgroup barrier, around a
// prologue and an epilogue as long as the loop itself -- 0.18 .. 0.22 of the f16 peak.  With 32 input channels the WHOLE K extent
// fits the LDS at once, so here nothing is left of that machinery:
//   * a persistent workgroup (one per CU, 4 waves) loads the layer's packed weights ONCE -- 9 steps x 4 KB = 36 KB, the same
//     A-fragment lane order as conv_split (rt_load.hip pack_weights_split with CC = 4: step s = tap s, lane group kb = cell kb) --
//     and keeps them for every tile it computes;
//   * an input tile is all 4 cells of (8 + 2) x (32 + 2 D) pixels, hi and lo planes: 45 .. 51 KB, double-buffered; the tile
//     after the current one is fetched by buffer-addressed LDS-DMA (out-of-image cells zero-filled by the range check) while
//     the current one is computed;
//   * the K loop is 216 MFMAs per wave of straight-line code (9 taps x 2 channel fragments x 4 pixel fragments x 3 products):
//     no DMA, no plan, no barrier inside it; ONE barrier per tile;
//   * epilogue as conv_split's: un-scale, bias, residual (centre-cropped), eval-BN affine, PReLU, split store, f16-range flag.
// Same tile geometry and launch windows as conv_split (rows strided by the dilation; SplitArgs::wy0 .. wx1), same tensor
// formats either side: the layers before and after do not know.  A pixel's sum runs over (tap, cell) in a fixed order whatever
// tile it falls into (internal tiling, windows: bit-identical); it is another order than conv_split's (chunk-major), so the two
// kernels agree to rounding, not to the bit.
// Measured (profiles/r05_rw_ab.txt, 4096^2): plain 1.02 - 1.16 ms per layer (276 - 316 TFLOP/s; conv_split: 1.44 - 1.71), with
// the residual 1.36 - 1.45 ms (226 - 236; 1.75 - 1.97).  That is 3.8 - 4.7 TB/s of algorithmic bytes (32 channels in, 32 out,
// 32 of residual at 4 B each against 18.4 kFLOP per pixel: 72 / 48 flop per byte) -- the layers are now HBM-bound, which is why
// a second form with TWO phase-shifted 4-wave groups per workgroup (one group's MFMAs under the other's epilogue and DMA issue;
// 8 x 16 tiles, four input buffers) measured 5 - 15 % SLOWER: more halo bytes per pixel for matrix-pipe time that was not the
// limit.  It is not kept.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_split.h"

namespace tpz {

// workgroup barrier that does NOT drain this wave's vector-memory operations: `__syncthreads()` is a fence + barrier, and the fence
// waits vmcnt(0) whenever an LDS-DMA (a pending LDS write on the VM counter) or a store is outstanding -- which here is always
// (the next tile's input, the previous tile's stores): every barrier then exposed a full memory latency.  What the barriers of
// these kernels order is LDS traffic only: each wave has waited for its OWN DMA pieces (vmcnt) before it arrives, and its LDS
// reads are complete (lgkmcnt) -- cdna_hip_programming.md, "Pipelining across barriers".
__device__ __forceinline__ void rw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D_>
struct RwCfg {
    static constexpr int K = 3, D = D_, MT = 32, TH = 8, TW = 32, CELLS = 4, WAVES = 4, THREADS = 256;
    static constexpr int MW = MT / 16, RPW = TH / WAVES, NFC = TW / 16, NW = RPW * NFC;
    static constexpr int ITH = TH + K - 1, ITW = TW + (K - 1) * D;
    static constexpr int CELL_STRIDE = (ITH * ITW + 15) / 16 * 16;      // cells per 8-channel cell plane (256-byte multiple: the
                                                                        // lane groups of a ds_read_b128 hit distinct banks)
    static constexpr int NPC = CELLS * CELL_STRIDE, PLANE_BYTES = NPC * 16, IN_BUF = 2 * PLANE_BYTES;
    static constexpr int NR = (NPC + THREADS - 1) / THREADS;            // DMA pieces per thread, plane and tile
    static constexpr int NSTEP = K * K, W_STEP_BYTES = 2 * MW * 1024, W_BYTES = NSTEP * W_STEP_BYTES;
    static constexpr int OFF_IN = W_BYTES, LDS_BYTES = W_BYTES + 2 * IN_BUF;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(PLANE_BYTES + ((RPW + K - 1) * ITW + TW + (K - 1) * D) * 16 < 65536, "ds_read immediates are 16 bit");
};

// uses of SplitArgs: in, wpk, wscale, bias, out, res (Hres, Wres, res_crop), post_scale / post_shift, flag, slope, Hin, Win,
// Hout, Wout, pad_x / pad_y, Hfull / Wfull (os = 1), wy0 .. wx1, tiles_x, tiles_y, n_tiles; cells_in = cells_out = 4
template <class C, int EPI>
__global__ __launch_bounds__(C::THREADS, 1) void conv_rw_kernel(const SplitArgs a) {
    constexpr int D = C::D, MW = C::MW, NW = C::NW, NFC = C::NFC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
    constexpr unsigned OOB = 0xfffffff0u;

    auto make_srd = [&](const void* base, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(base)), 0,
                                                 (int)__builtin_amdgcn_readfirstlane((unsigned)bytes), 0x00020000);
    };
    auto bdma16 = [&](__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned lds_addr) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(size_t)lds_addr, 16, voff, 0, 0, 0);
    };

    // ---- this workgroup's tiles: the tiles are dealt to the XCDs in eight contiguous runs (block b runs on XCD b % 8:
    // neighbouring tiles share their halo in that XCD's L2), round-robin to the XCD's workgroups within a run
    const unsigned nt = (unsigned)a.n_tiles;
    unsigned tile_L, tile_end, tile_stride;
    {
        const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, q = nt / 8, r = nt % 8;
        const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_end = start + q + (xcd < r ? 1u : 0u);
        tile_stride = gridDim.x >> 3;
        tile_L = start + j;
    }
    if (tile_L >= tile_end) return;

    // ---- the layer's weights: 36 KB, once
    {
        const __amdgpu_buffer_rsrc_t srd_w = make_srd(a.wpk, (size_t)C::W_BYTES);
#pragma unroll
        for (int i = 0; i < C::W_BYTES / (C::THREADS * 16); ++i)
            bdma16(srd_w, (unsigned)(i * C::THREADS + tid) * 16u,
                   __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((i * C::THREADS + wave * 64) * 16)));
    }

    // ---- the pieces of an input tile this thread fetches: piece g = r * THREADS + tid is cell c, tile row rr, tile column xx;
    // its offset relative to the tile origin is tile-invariant, its position decides per tile whether it lies inside the image
    const size_t plane_bytes = (size_t)C::CELLS * a.Hin * a.Win * 16;          // hi plane (all cells); lo follows
    const __amdgpu_buffer_rsrc_t srd_hi = make_srd(a.in, plane_bytes);
    const __amdgpu_buffer_rsrc_t srd_lo = make_srd(reinterpret_cast<const unsigned char*>(a.in) + plane_bytes, plane_bytes);
    unsigned rel[C::NR], pos[C::NR];
    unsigned exists_mask = 0;              // bit r: piece r of this thread is a cell of the tile (not padding of the cell plane)
    static_assert(C::NR <= 32, "exists_mask");
#pragma unroll
    for (int r = 0; r < C::NR; ++r) {
        const int g = r * C::THREADS + tid;
        const int c = g / C::CELL_STRIDE, rem = g - c * C::CELL_STRIDE;
        const int rr = rem / C::ITW, xx = rem - rr * C::ITW;
        const bool exists = g < C::NPC && rem < C::ITH * C::ITW;
        rel[r] = (unsigned)(((size_t)c * a.Hin + (size_t)rr * D) * a.Win + xx) * 16u;
        pos[r] = (unsigned)(rr * D) << 16 | (unsigned)xx;
        exists_mask |= exists ? 1u << r : 0u;       // (a predicate of its own: no sentinel position that a tall image could reach)
    }
    int y0, x0;
    auto set_tile = [&](unsigned L) {
        const unsigned by = L / (unsigned)a.tiles_x, bx = L - by * (unsigned)a.tiles_x;
        y0 = a.wy0 + (int)(by / D) * (C::TH * D) + (int)(by % D);
        x0 = a.wx0 + (int)bx * C::TW;
    };
    auto fetch = [&](int ty0, int tx0, int buf) {
        const int ybase = ty0 - a.pad_y, xbase = tx0 - a.pad_x;
        const unsigned base = (unsigned)((ybase * a.Win + xbase) * 16);     // (mod 2^32: exact whenever the piece is inside)
#pragma unroll
        for (int r = 0; r < C::NR; ++r) {
            if ((r + 1) * C::THREADS <= C::NPC || r * C::THREADS + wave * 64 < C::NPC) {     // (whole waves of 1 KiB)
                const int gy = ybase + (int)(pos[r] >> 16), gx = xbase + (int)(pos[r] & 0xffffu);
                const bool in = ((exists_mask >> r) & 1u) && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
                const unsigned off = in ? rel[r] + base : OOB;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(C::OFF_IN + buf * C::IN_BUF + (r * C::THREADS + wave * 64) * 16));
                bdma16(srd_hi, off, dst);
                bdma16(srd_lo, off, dst + C::PLANE_BYTES);
            }
        }
    };

    set_tile(tile_L);
    fetch(y0, x0, 0);
    const unsigned b_lane = (unsigned)(C::OFF_IN + ((l4 * C::CELL_STRIDE) + (wave * C::RPW) * C::ITW + l15) * 16);
    const unsigned a_lane = (unsigned)(lane * 16);
    auto b_off = [](int n) { return ((n / NFC) * C::ITW + (n % NFC) * 16) * 16; };
    // ---- epilogue addressing: buffer accesses with 32-bit lane offsets (hi plane; lo = its own descriptor), OOB for pixels
    // outside the launch window -- the hardware drops those stores; residual loads use clamped pixels (always in range)
    const unsigned cp16_o = (unsigned)((size_t)a.Hfull * a.Wfull * 16), cp16_r = (unsigned)((size_t)a.Hres * a.Wres * 16);
    const size_t pl16_o = (size_t)a.cells_out * cp16_o, pl16_r = (size_t)a.cells_out * cp16_r;
    const __amdgpu_buffer_rsrc_t srd_oh = make_srd(a.out, pl16_o);
    const __amdgpu_buffer_rsrc_t srd_ol = make_srd(reinterpret_cast<const unsigned char*>(a.out) + pl16_o, pl16_o);
    const bool has_res = (EPI == EPI_RES || EPI == EPI_RES_POST);
    const __amdgpu_buffer_rsrc_t srd_rh = make_srd(has_res ? (const void*)a.res : (const void*)a.out, has_res ? pl16_r : 16);
    const __amdgpu_buffer_rsrc_t srd_rl =
        make_srd(has_res ? (const void*)(reinterpret_cast<const unsigned char*>(a.res) + pl16_r) : (const void*)a.out, has_res ? pl16_r : 16);
    // (lane part of an access: (second cell of the channel fragment) cell planes; fragment m adds 2 m cell planes)
    const f32x2 slope2 = {a.slope, a.slope};
    bool big = false;
    int buf = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the weights and this workgroup's first tile
    for (;;) {
        const bool has_next = tile_L + tile_stride < tile_end;
        const int cy0 = y0, cx0 = x0;
        // every wave's pieces of the current tile are in the LDS (each waited for its own behind its previous K loop) and every
        // wave is done reading the other buffer
        rw_barrier();
        // per PAIR of pixel fragments (2q, 2q + 1: the two 16-pixel halves of tile row q of this wave): output / residual offsets of
        // this lane (window test folded into the offset).  As in conv_split (round 6): a v_permlane16_swap per dword leaves the
        // lanes of an even 16-lane row with both halves of their cell of fragment 2q's pixel and the lanes of an odd row with both
        // halves of the same cell of fragment 2q + 1's pixel, so every store and every residual load is 16 bytes per lane -- half the
        // memory instructions of the 8-byte form, every byte where it went before.
        static_assert(NW % 2 == 0 && NFC == 2, "fragment pairs = the two halves of a tile row");
        constexpr int NP = NW / 2;
        unsigned ovw[NP], rvw[NP];
        bool okp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int oy = cy0 + (wave * C::RPW + q) * D;
            const int ox = cx0 + (l4 & 1) * 16 + l15;
            const bool ok = oy < a.wy1 && ox < a.wx1;
            const int ry = oy < a.Hout ? oy : a.Hout - 1, rx = ox < a.Wout ? ox : a.Wout - 1;      // (clamped: loads stay in range)
            okp[q] = ok;
            ovw[q] = ok ? (unsigned)(oy * a.Wfull + ox) * 16u + (unsigned)(l4 >> 1) * cp16_o : OOB;
            rvw[q] = (unsigned)((ry + a.res_crop) * a.Wres + rx + a.res_crop) * 16u + (unsigned)(l4 >> 1) * cp16_r;
        }
        // the residual cells of the tile are requested NOW and used behind the K loop: their latency hides under the MFMAs
        u32x4 rch[MW][NP], rcl[MW][NP];
        if constexpr (EPI == EPI_RES || EPI == EPI_RES_POST) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    rch[m][q] = __builtin_amdgcn_raw_buffer_load_b128(srd_rh, (int)(rvw[q] + 2u * m * cp16_r), 0, 0);
                    rcl[m][q] = __builtin_amdgcn_raw_buffer_load_b128(srd_rl, (int)(rvw[q] + 2u * m * cp16_r), 0, 0);
                }
        }
        if (has_next) {
            set_tile(tile_L + tile_stride);
            fetch(y0, x0, buf ^ 1);                      // in flight under this tile's MFMAs
        }
        // ---- K loop: 9 taps, all 4 cells of a tap per step (lane group kb = cell)
        f32x4 acc[MW][NW];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned char* bl = lds + b_lane + buf * C::IN_BUF;
        f16x8 bh[2][NW], bo[2][NW], ah[2][MW], ao[2][MW];
        auto load_step = [&](int s, int slot) {
            const int ky = s / C::K, kx = s - ky * C::K;
            const int tap = (ky * C::ITW + kx * D) * 16;
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                bh[slot][n] = *reinterpret_cast<const f16x8*>(bl + tap + b_off(n));
                bo[slot][n] = *reinterpret_cast<const f16x8*>(bl + tap + b_off(n) + C::PLANE_BYTES);
            }
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                ah[slot][m] = *reinterpret_cast<const f16x8*>(lds + a_lane + s * C::W_STEP_BYTES + m * 1024);
                ao[slot][m] = *reinterpret_cast<const f16x8*>(lds + a_lane + s * C::W_STEP_BYTES + (MW + m) * 1024);
            }
        };
        load_step(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NW + 2 * MW, 0);
#pragma unroll
        for (int s = 0; s < C::NSTEP; ++s) {
            const int cur = s & 1;
            // issue order pinned: the fragment reads of step s + 1, then the MFMAs of step s (one wave per SIMD: nothing else
            // hides an LDS round trip)
            if (s + 1 < C::NSTEP) {
                load_step(s + 1, cur ^ 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NW + 2 * MW, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * MW * NW, 0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bo[cur][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int n = 0; n < NW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ao[cur][m], bh[cur][n], acc[m][n], 0, 0, 0);
        }
        // the next tile's pieces (issued a K loop ago), the residual cells: landed.  The stores below are NOT waited for here --
        // they drain under the next tile's K loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- epilogue: lane (l15, l4) holds channels m * 16 + 4 * l4 .. + 3 of pixel l15 of fragment n
        u16x2 bigacc = {0, 0};
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int co0 = m * 16 + l4 * 4;
            f32x2 sc[2], bi[2], psc[2], psh[2];
            {
                const float4 s4 = *reinterpret_cast<const float4*>(a.wscale + co0);
                sc[0] = (f32x2){s4.x, s4.y}; sc[1] = (f32x2){s4.z, s4.w};
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) b4 = *reinterpret_cast<const float4*>(a.bias + co0);
                bi[0] = (f32x2){b4.x, b4.y}; bi[1] = (f32x2){b4.z, b4.w};
                if constexpr (EPI == EPI_RES_POST) {
                    const float4 p4 = *reinterpret_cast<const float4*>(a.post_scale + co0);
                    const float4 q4 = *reinterpret_cast<const float4*>(a.post_shift + co0);
                    psc[0] = (f32x2){p4.x, p4.y}; psc[1] = (f32x2){p4.z, p4.w};
                    psh[0] = (f32x2){q4.x, q4.y}; psh[1] = (f32x2){q4.z, q4.w};
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                // the residual cells of the pair -> half cells in accumulator layout (the swap is an involution)
                u32x2 rh2[2] = {{0u, 0u}, {0u, 0u}}, rl2[2] = {{0u, 0u}, {0u, 0u}};
                if constexpr (EPI == EPI_RES || EPI == EPI_RES_POST) {
                    const auto h0s = __builtin_amdgcn_permlane16_swap(rch[m][q][0], rch[m][q][2], false, false);
                    const auto h1s = __builtin_amdgcn_permlane16_swap(rch[m][q][1], rch[m][q][3], false, false);
                    const auto l0s = __builtin_amdgcn_permlane16_swap(rcl[m][q][0], rcl[m][q][2], false, false);
                    const auto l1s = __builtin_amdgcn_permlane16_swap(rcl[m][q][1], rcl[m][q][3], false, false);
                    rh2[0] = (u32x2){h0s[0], h1s[0]}; rh2[1] = (u32x2){h0s[1], h1s[1]};
                    rl2[0] = (u32x2){l0s[0], l1s[0]}; rl2[1] = (u32x2){l0s[1], l1s[1]};
                }
                unsigned hh[2][2], ll[2][2];          // [fragment of the pair][dword]
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = 2 * q + j;
                    f32x2 v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 addend = bi[h];
                        if constexpr (EPI == EPI_RES || EPI == EPI_RES_POST)
                            addend = add_halves(add_halves(addend, h ? rl2[j].y : rl2[j].x), h ? rh2[j].y : rh2[j].x);
                        v[h] = (f32x2){acc[m][n][2 * h], acc[m][n][2 * h + 1]} * sc[h] + addend;
                        if constexpr (EPI == EPI_RES_POST) v[h] = v[h] * psc[h] + psh[h];
                        v[h] = __builtin_elementwise_max(v[h], v[h] * slope2);
                    }
                    split2m(v[0], hh[j][0], ll[j][0]);
                    split2m(v[1], hh[j][1], ll[j][1]);
                }
                // f16-range check on what is stored: after the swap a lane holds its OWN pixel's cell (fragment l4 & 1 of the pair)
                const auto sh0 = __builtin_amdgcn_permlane16_swap(hh[0][0], hh[1][0], false, false);
                const auto sh1 = __builtin_amdgcn_permlane16_swap(hh[0][1], hh[1][1], false, false);
                const auto sl0 = __builtin_amdgcn_permlane16_swap(ll[0][0], ll[1][0], false, false);
                const auto sl1 = __builtin_amdgcn_permlane16_swap(ll[0][1], ll[1][1], false, false);
                const u32x4 chi = {sh0[0], sh1[0], sh0[1], sh1[1]}, clo = {sl0[0], sl1[0], sl0[1], sl1[1]};
                const unsigned okmask = okp[q] ? 0x7fff7fffu : 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) bigacc = __builtin_elementwise_max(bigacc, __builtin_bit_cast(u16x2, chi[k] & okmask));
                const unsigned off = okp[q] ? ovw[q] + 2u * m * cp16_o : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(chi, srd_oh, (int)off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(clo, srd_ol, (int)off, 0, 0);
            }
        }
        big = big || bigacc[0] >= 0x7c00 || bigacc[1] >= 0x7c00;
        if (!has_next) break;
        tile_L += tile_stride;
        buf ^= 1;
    }
    if (__any(big) && lane == 0) atomicOr(a.flag, 1u);
}

// launch: grid = workgroups (a multiple of 8), a.n_tiles = tiles_x * tiles_y
template <int DIL, int EPI>
hipError_t launch_rw_cfg(const SplitArgs& a, int workgroups, hipStream_t s) {
    using C = RwCfg<DIL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rw_kernel<C, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_rw_kernel<C, EPI>), dim3((unsigned)workgroups), dim3(C::THREADS), C::LDS_BYTES, s, a);
    return hipGetLastError();
}

// dil in {1, 2, 4}, epi in {EPI_PLAIN, EPI_RES, EPI_RES_POST}; hipErrorInvalidValue otherwise
hipError_t launch_conv_rw(const SplitArgs& a, int dil, int epi, int workgroups, hipStream_t s);

}  // namespace tpz
