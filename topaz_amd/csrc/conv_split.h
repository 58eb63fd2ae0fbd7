// fp32-accurate implicit-GEMM convolution on the f16 matrix cores of CDNA4 (v_mfma_f32_16x16x32_f16).
//
// The dilated convolutions of the filled scoring networks (topaz/model/features/resnet.py:129-133,185-202,
// 294-302; topaz/model/features/basic.py:47-63) dominate `topaz extract`; on the fp32 MFMA path
// (conv_mfma.h) they sit at ~93 % of a 157 TFLOP/s peak.  The f16 MFMA issues 16x the MACs per cycle, so
// every fp32 operand is carried as two f16 halves (split_fmt.h) and each product is formed as
//     w*x  ~=  wh*xh + wh*xl + wl*xh          (the dropped wl*xl term is < 2^-22 |w*x|)
// -- three exact f16 products accumulated in fp32: 16/3 = 5.3x the fp32-MFMA rate at fp32-LEVEL accuracy
// (the split carries 22 mantissa bits, fp32 24).  Measured against a float64 evaluation: one layer within 1.5x of
// torch's fp32 convolution (tests/test_gpu_split.py), the benchmark's networks end to end 1.1 - 1.7x torch's own fp32
// error and far inside the 1e-4 bar (tests/test_gpu_precision.py) -- not below it.  Weights are scaled per
// output channel by a power of two before splitting (undone exactly in the epilogue) so that their lo
// halves stay normal f16 numbers.  Activations beyond the f16 range raise a device flag; the runtime then
// re-runs the forward pass on the fp32 kernels (rt_forward.hip, rt_denoise.hip).
//
// GEMM view: M = output channels, N = output pixels, K = Cin * taps.  One MFMA covers 32 K-elements:
//   lane (i, kb) holds 8 consecutive K values; kb = 0..3 selects a "slot" = (tap, 8-channel cell), the 8
//   values are the cell's channels.  B operand = one 16-byte cell of the LDS tile (ds_read_b128), A operand
//   = 16 bytes of host-packed weights.  A chunk of the K loop is CC cells (8*CC channels) x all taps, cut
//   into steps of 4 slots (SplitCfg::slot): with CC even the slot pairs (kb 0,1) and (kb 2,3) are the two
//   cells of one tap; with CC = 1 they are vertically adjacent taps.  Either way the two 16-byte reads of a
//   ds_read_b128 lane group are a multiple of 256 bytes apart: conflict-free (MI355X_MICROARCH.md, LDS).
// Workgroup = 512 threads = 8 waves (two per SIMD), one per CU -- or, where the LDS tile fits 80 KB (dilation 1,
//   Cout <= 96), 256 threads = 4 waves, two workgroups per CU so that one's prologue / epilogue / barrier waits
//   hide under the other's MFMAs; tile = MT output channels x (TH x TW) pixels, rows strided by the dilation as
//   in conv_mfma.h.  Wave tile = MT x (TH/WAVES rows x TW) pixels: per step 2*MW + 2*NW ds_read_b128 feed
//   3*MW*NW MFMAs.
// The same kernel carries the U-Nets (denoising/models.py:74-244, 452-564): a second source with the nearest
//   upsample + concat folded into the loader, per-axis pads and a strided output lattice (the per-parity forms
//   of the decoder convs, SplitArgs::nphase / subpix_cout), K x 1 "column" tap shapes (stems, 1-output-channel
//   convs), an fp32-storing and a max-pooling epilogue.
// Pipeline: one stage = SPS steps.  Weights of the next stage and, spread over the steps of a chunk, the input
//   tile of the next chunk arrive by LDS-DMA (buffer_load_dwordx4 ... lds: one cell per lane, out-of-image lanes
//   zero-filled by the range check) into the other LDS buffers while the stage computes; one barrier per stage.
//   What each step fetches and where its B fragments lie is a host-built table (SplitStep) read through the scalar
//   cache; the kernel is compiled per addressing mode (MODE: single source / everything / persistent workgroups).
// 3-D convolutions (UDenoiseNet3D, denoising/models.py:452-564; the 3-D scoring networks, classifier.py:69-102)
//   run on the same 2-D tiles: output plane z of a k^3 conv of dilation D is the 2-D k^2 conv of the k input planes
//   z + kz*D - pad stacked as channels, so the K loop walks
//   "virtual cells" v = kz * cells + c (SplitArgs::KZ); one grid-z slice per output plane.  Every input plane is
//   read by k output planes (a 3-D LDS tile's halo re-reads as much), the K loop is k times longer (good).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "conv_mfma.h"
#include "split_fmt.h"

namespace tpz {

struct SplitArgs {
    const uint4* in;          // split tensor: [2 planes][cells_in1][H1][W1] cells
    const uint4* in2;         // optional second source [2][cells_in - cells_in1][Hin][Win]: channels after those of
                              // `in`, which is then nearest-upsampled to Hin x Win (fused upsample + concat)
    const uint4* wpk;         // packed weights (rt_load.hip pack_weights_split)
    const float* wscale;      // [Cout] 2^-s: undoes the per-channel weight scaling.  This and the other per-channel vectors
                              // are zero-padded to whole tiles (rt_internal.h chan_pad): fetched as unclamped float4
    const float* bias;        // [Cout] or nullptr
    uint4* out;               // split tensor [2][cells_out][Hfull][Wfull] (nullptr: fused head / fp32 output)
    float* out_f32;           // EPI_PLAIN_F32: fp32 planes [Cout][Hfull][Wfull] instead
    const uint4* res;         // split residual [2][cells_out][Hres][Wres] or nullptr (may alias out: in place)
    const float* post_scale;  // [Cout] affine after the residual add (eval BN) or nullptr
    const float* post_shift;
    const float* head_w;      // fused 1x1 head: [Cout]
    float* head_out;          // fp32 [Hout][Wout]
    const void* zeros;        // >= 16 bytes of zeros (source of padded / outside cells)
    unsigned* flag;           // set to 1 when a stored activation leaves the f16 range
    float head_b, slope;
    int cells_in, Hin, Win;   // all input cells; logical input geometry
    int cells_in1, H1, W1;    // cells and geometry of `in` (== cells_in, Hin, Win without a second source)
    int Cout, cells_out, Hout, Wout;
    int pad_x, pad_y;
    // output lattice as in conv_mfma.h: element (oy, ox) of the launch is element (oy*os + ooy, ox*os + oox)
    // of the [Hfull][Wfull] output; the residual is read at the same full-tensor position (+ res_crop)
    int os, ooy, oox, Hfull, Wfull;
    int Hres, Wres, res_crop;
    // 3-D mode (KZ >= 1 with tensors [cells][D][H][W]; 2-D: KZ = 1, every D = 1, ncz = grid z); `in2` must have the
    // geometry of `in` there (no upsampling).
    int KZ, pad_z, Din, Dout, ooz, Dfull, Dres;
    int ncz;                  // co-group slices of grid z: blockIdx.z = (plane * max(nphase, 1) + phase) * ncz + slice
    // all output parities of a decoder conv in one launch (rt_load.hip prepare_split_phases): phase p = (pz, py, px)
    // bits uses weights wpk + p * w_phase_bytes, scales wscale + p * Cout, pad (phase_k/2 - parity + 1)/2 and
    // lattice offset = parity on each axis (pad_* / oo* of the struct are then ignored)
    int nphase, phase_k, ws_phase_stride;
    size_t w_phase_bytes;
    // or, when every parity reads the same window (5x5 -> 3x3 taps, pad 1): the parities as 4 * Cout VIRTUAL output
    // channels of one conv (sub-pixel convolution), virtual channel v = parity * Cout + co stored at pixel
    // (2*oy + py, 2*ox + px), channel co.  subpix_cout = the real Cout (0 = off); wscale has 4 * Cout entries.
    int subpix_cout;
    int issuer_half;          // 1: only the upper half of the waves issues the per-step DMA (each for two waves' shares)
    int n_chunks;             // chunks of CC (virtual) cells
    int cog_inner;            // co-groups looped inside the kernel (fused head), else 1
    int tiles_x, tiles_y;
    int n_tiles;              // persistent launches (MODE 4): tiles_x * tiles_y * grid-z slices, walked by gridDim.x workgroups
    int vol_srcmajor;         // plane-stacked 3-D with two sources: virtual cells ordered source-major (all planes of `in`, then all
                              // planes of `in2`) -- no chunk mixes the two tensors (MODE 11); 0: plane-major (MODE 3)
    int xcd_swizzle;
    // output WINDOW of the launch, in its own lattice coordinates: the tiles cover [wy0, wy1) x [wx0, wx1) only and nothing
    // outside it is stored (launch_split defaults it to the whole lattice).  A patch of a patched denoise keeps only its
    // centre, so every layer computes only the part of its tensor that the kept pixels depend on (rt_exec.hip, need_regions).
    int wy0, wx0, wy1, wx1;
    // ... and, plane-stacked 3-D, the planes [wz0, wz0 + Dout) of the launch lattice (a tile of a tiled tomogram keeps its centre
    // in z as well).  Dlat: the lattice's full depth (host side only: the share of the layer's FLOP a windowed launch executes)
    int wz0, Dlat;
    // FOLDED 1x1 projection (ResidA: y = conv1(t) + proj(h), resnet.py:185-202): the last fold_cells cells of the K loop come
    // from `in2` = h with ONE tap each -- the centre tap of the tile layout, read at (row + in2_oy, column + in2_ox) of an
    // in2_H x in2_W tensor -- instead of a separate 1x1 pass that writes a 128-channel residual tensor and reads it back
    // (13 GB per micrograph).  In the slot stream each folded chunk (CC = 2 cells) is one step: slots (centre tap, cell 0 / 1)
    // and (next tap, cell 0 / 1), the latter with zero weights.  Kernels with one step per stage only (SPS == 1).
    int fold_cells, fold_tap, in2_H, in2_W, in2_oy, in2_ox;
    // the tile-invariant schedule of the K loop (split_make_plan, built once per layer on the host): entry 0 describes the
    // fetch of chunk 0 in the prologue, entry 1 + s step s -- where its B fragments lie in the LDS and which share of which
    // chunk's input tile it prefetches.  The kernel reads one entry per step through the scalar cache.
    const struct SplitStep* plan;
};

// One step of a tile's K loop as the kernel sees it (32 bytes = one s_load_dwordx8).  Everything that depends only on the
// layer (cells, taps, sources, stages) and not on the tile is decided here, on the host: the kernel's own per-step scalar
// code is a handful of instructions (round 2: ~350 instructions with 25 branches per step on the 128-channel tiles,
// executed by all eight waves before their first MFMA of the step).
struct SplitStep {
    uint16_t bofs[4];         // LDS byte offset / 16 of the step's B fragments for lane group kb: input buffer + slot (tap, cell)
    uint32_t dma;             // input DMA issued at this step (SPLIT_DMA_*)
    uint32_t cidx;            // 2-D: index of the fetched chunk within its source tensor
    uint32_t cell[4];         // plane-stacked 3-D: per cell of the fetched chunk  bit 0 valid, bit 1 from `in2`,
                              // bits 2-7 kz, bits 8.. cell index within its source
};
enum : uint32_t {
    SPLIT_DMA_ROUNDS = 0xffu,         // bit r: DMA round r of the chunk's tile is issued at this step
    SPLIT_DMA_BUF = 1u << 8,          // destination input buffer
    SPLIT_DMA_NCELL_SHIFT = 9,        // bits 9-11: the chunk's cells that exist when fewer than CC (0 = all CC)
    SPLIT_DMA_SWITCH = 1u << 12,      // first fetch from the second source: its offset table replaces the first one's
    SPLIT_DMA_SRC2 = 1u << 13,        // 2-D: the chunk comes from `in2`
    SPLIT_DMA_NEXT = 1u << 14,        // the fetch is chunk 0 of the workgroup's NEXT tile (persistent kernels; others skip it);
                                      // with SPLIT_DMA_SWITCH: its first round -- the offset table becomes the next tile's
    SPLIT_DMA_ANY = 1u << 31,
};

struct SplitSlot {
    int ky, kx, c;            // ky < 0: padding slot (zero weights)
};

// K x KX taps (rows x columns; KX = K for the square kernels, KX = 1 for the column kernels that carry the x taps of
// a stem as input channels or of a 1-output-channel conv as output channels, rt_load.hip prepare_split)
template <int K_, int D_, int MT_, int TH_, int TW_, int CC_, int WAVES_ = 8, int KX_ = K_, int SPS_ = 1>
struct SplitCfg {
    static constexpr int K = K_, KX = KX_, D = D_, MT = MT_, TH = TH_, TW = TW_, CC = CC_;
    // 8 waves: one workgroup per CU (the big dilated tiles need most of the LDS); 4 waves: two per CU, so that
    // one workgroup's prologue / epilogue / barrier waits overlap the other's MFMAs (small-halo, short-K layers)
    static constexpr int WAVES = WAVES_, THREADS = 64 * WAVES_, WGS_PER_CU = WAVES_ == 8 ? 1 : 2;
    static constexpr bool UNIFORM_DMA = (MT_ >= 128 && WAVES_ == 8);    // how the LDS-DMA is issued (see fetch)
    // the upper half of an 8-wave workgroup may issue the whole step's DMA (SplitArgs::issuer_half): -3 .. -4 % on the wide tiles
    static constexpr bool ISSUER_HALF = (MT_ >= 96 && WAVES_ == 8);
    static constexpr int MW = MT / 16;
    static constexpr int RPW = TH / WAVES;
    static constexpr int NFC = TW / 16;
    static constexpr int NW = RPW * NFC;
    static constexpr int ITH = TH + K - 1;
    // the two 16-byte reads of a ds_read_b128 lane group must be a multiple of 256 B apart: vertically adjacent
    // taps (CC = 1) need rows of a multiple of 16 cells, the two cells of one tap (CC even) only a cell plane of one
    static constexpr int ITW = (CC == 1) ? (TW + (KX - 1) * D + 15) / 16 * 16 : TW + (KX - 1) * D;
    static constexpr int CELL_STRIDE = (ITH * ITW + 15) / 16 * 16;  // cells per 8-channel cell plane
    static constexpr int NPC = CC * CELL_STRIDE;                    // cells per (hi | lo) plane of a chunk
    static constexpr int PLANE_BYTES = NPC * 16;
    static constexpr int IN_BUF = 2 * PLANE_BYTES;
    static constexpr int NR = (NPC + THREADS - 1) / THREADS;        // DMA rounds per plane
    // ---- slots
    static constexpr bool ROWPAIR = (K == 5) && (KX == 5) && ((4 * D) % 16 == 0);   // last-row taps kx = 0 and 4 pair up
    static constexpr int PV = KX * (K / 2);                             // vertical tap pairs (CC == 1)
    static constexpr int PAIRS1 = PV + ((K % 2) ? (ROWPAIR ? KX - 1 : KX) : 0);
    static constexpr int NSTEP = (CC == 1) ? (PAIRS1 + 1) / 2 : (K * KX * CC + 3) / 4;
    // CC even: the slots of consecutive chunks form ONE stream, q = G % Q, chunk = G / Q for the global slot
    // G = 4 * stage + kb -- a step may take its first two slots from the end of chunk c and the last two from the
    // start of chunk c + 1 (both tiles are resident: double buffer), so only the very last step of a tile is padded:
    // 3x3 x 2 cells = 18 slots per chunk cost 4.5 steps instead of 5, 5x5 x 2 = 50 slots 12.5 instead of 13.
    static constexpr bool CONT = (CC % 2 == 0);
    static constexpr int Q = K * KX * CC;                               // slots per chunk (CONT)
    // `cells` (virtual) cells in all: the last chunk may hold fewer than CC -- its slots enumerate its own cells only
    static constexpr int TAPS = K * KX;
    __host__ __device__ static constexpr int cont_slots(int cells) { return TAPS * cells; }
    __host__ __device__ static constexpr int cont_stages(int cells) { return (TAPS * cells + 3) / 4; }
    __host__ __device__ static constexpr SplitSlot cont_slot(int q) { return SplitSlot{(q / CC) / KX, (q / CC) % KX, q % CC}; }
    __host__ __device__ static constexpr SplitSlot slot(int step, int kb) {
        if (CC != 1) {
            const int q = step * 4 + kb, t = q / CC;
            return t < K * KX ? SplitSlot{t / KX, t % KX, q % CC} : SplitSlot{-1, 0, 0};
        }
        const int p = step * 2 + (kb >> 1), j = kb & 1;
        if (p < PV) return SplitSlot{2 * (p % (K / 2)) + j, p / (K / 2), 0};
        const int e = p - PV;
        if (K % 2 == 0) return SplitSlot{-1, 0, 0};
        if (ROWPAIR) {
            if (e == 0) return SplitSlot{K - 1, j ? 4 : 0, 0};
            return (e < KX - 1 && j == 0) ? SplitSlot{K - 1, e, 0} : SplitSlot{-1, 0, 0};
        }
        return (e < KX && j == 0) ? SplitSlot{K - 1, e, 0} : SplitSlot{-1, 0, 0};
    }
    __host__ __device__ static constexpr int slot_lds_off(int step, int kb) {
        SplitSlot s = slot(step, kb);
        if (s.ky < 0) s = slot(step, kb & ~1);          // padding reads its partner's cell (weights are zero)
        if (s.ky < 0) return 0;
        return (s.c * CELL_STRIDE + s.ky * ITW + s.kx * D) * 16;
    }
    static constexpr int W_STEP_BYTES = 2 * MW * 1024;              // hi + lo A fragments of one step
    // A STAGE is SPS consecutive steps between two workgroup barriers: its weights (SPS * W_STEP_BYTES, contiguous in the
    // packed array) arrive by DMA during the previous stage.  Narrow channel tiles (MT <= 64: 24 - 48 MFMAs per wave and
    // step) would otherwise synchronise every 400 - 800 cycles.
    static constexpr int SPS = SPS_;
    static constexpr int W_STAGE_BYTES = SPS * W_STEP_BYTES;
    static constexpr int WR = (W_STAGE_BYTES + THREADS * 16 - 1) / (THREADS * 16);  // DMA rounds per weight stage
    static constexpr int OFF_W = 2 * IN_BUF;
    static constexpr int OFF_TAB = OFF_W + 2 * W_STAGE_BYTES;
    static constexpr int LDS_BYTES = OFF_TAB + (NPC * 4 + 15) / 16 * 16;
    static_assert(NR <= 8, "the plan carries the DMA rounds of a step as a byte mask");
    static_assert(TH % WAVES == 0 && TW % 16 == 0 && MT % 16 == 0, "tile shape");
    static_assert(PLANE_BYTES + (RPW * ITW + TW) * 16 < 65536, "ds_read immediates are 16 bit");
    static_assert(LDS_BYTES <= 160 * 1024 / WGS_PER_CU, "LDS per workgroup");
    static_assert(WAVES == 8 || WAVES == 4, "waves per workgroup");
    // every chunk's prefetch window (~ Q / 4 - 1 steps) must contain at least one whole, aligned stage
    static_assert(SPS == 1 || (CONT && Q >= 8 * SPS), "multi-step stages: continuous slot stream, chunks of >= 2 * SPS steps");
};

// What a plan depends on (besides the kernel configuration): the K loop's cells and sources, nothing of the tile or image size
struct SplitPlanKey {
    int cells_in, cells_in1, n_chunks, has_in2, vol, KZ, fold_cells, fold_tap, srcmajor;
};

// Host: the schedule of one tile's K loop (SplitStep) for configuration C.  Entry 0 = the fetch of chunk 0 (all rounds, issued
// in the prologue), entry 1 + s = step s.  Mirrors pack_weights_split (rt_load.hip), which lays the weights out in the same
// (step, lane group) -> (chunk, tap, cell) order.
template <class C>
void split_make_plan(const SplitPlanKey& k, std::vector<SplitStep>& out) {
    const bool vol = k.vol != 0;
    const bool folded = C::CONT && C::SPS == 1 && k.fold_cells > 0;
    const int vcells = (folded ? k.cells_in1 : k.cells_in) * (vol ? k.KZ : 1);
    const int n_full = vcells / C::CC, n_rem = vcells - n_full * C::CC;
    const int n_stages_a = C::CONT ? C::cont_stages(vcells) : k.n_chunks * C::NSTEP;
    const int n_stages = n_stages_a + (folded ? k.fold_cells / C::CC : 0);
    const int chunks1 = (k.has_in2 && !vol) ? k.cells_in1 / C::CC : k.n_chunks;
    auto cont_off = [](int q) {
        const SplitSlot sl = C::cont_slot(q);
        return (sl.c * C::CELL_STRIDE + sl.ky * C::ITW + sl.kx * C::D) * 16;
    };
    // the fetch of chunk `pf`, rounds `rmask`
    auto fetch = [&](SplitStep& e, int pf, unsigned rmask, bool sw) {
        e.dma = SPLIT_DMA_ANY | (rmask & SPLIT_DMA_ROUNDS) | ((pf & 1) ? SPLIT_DMA_BUF : 0u) | (sw ? SPLIT_DMA_SWITCH : 0u);
        if (!vol) {
            const bool second = pf >= chunks1;
            if (second) e.dma |= SPLIT_DMA_SRC2;
            e.cidx = (uint32_t)(second ? pf - chunks1 : pf);
            const int have = k.cells_in - pf * C::CC;
            if (have < C::CC) e.dma |= (uint32_t)(have > 1 ? have : 1) << SPLIT_DMA_NCELL_SHIFT;     // (a chunk that exists has >= 1 cell)
        } else {
            bool any2 = false, any1 = false;
            for (int j = 0; j < C::CC && j < 4; ++j) {
                const int v = pf * C::CC + j;
                int kz, c;                     // plane and cell (counted over both sources) of virtual cell v
                if (!k.srcmajor) {
                    kz = v / k.cells_in; c = v - kz * k.cells_in;
                } else if (v < k.KZ * k.cells_in1) {
                    kz = v / k.cells_in1; c = v - kz * k.cells_in1;
                } else {
                    const int vv = v - k.KZ * k.cells_in1, c2n = k.cells_in - k.cells_in1;
                    kz = vv / c2n; c = k.cells_in1 + (vv - kz * c2n);
                }
                if (kz >= k.KZ) { e.cell[j] = 0; continue; }
                const bool l2 = c >= k.cells_in1;
                (l2 ? any2 : any1) = true;
                e.cell[j] = 1u | (l2 ? 2u : 0u) | (uint32_t)kz << 2 | (uint32_t)(l2 ? c - k.cells_in1 : c) << 8;
            }
            if (k.srcmajor && any2 && !any1) e.dma |= SPLIT_DMA_SRC2;      // (the host orders source-major only when no chunk mixes)
        }
    };
    out.assign((size_t)n_stages + 1, SplitStep{});
    fetch(out[0], 0, (1u << C::NR) - 1, false);
    unsigned next_rounds = 0;
    for (int s = 0; s < n_stages; ++s) {
        SplitStep& e = out[(size_t)s + 1];
        for (int l4 = 0; l4 < 4; ++l4) {
            int off;
            if (C::CONT) {
                if (folded && s >= n_stages_a) {
                    off = ((chunks1 + s - n_stages_a) & 1) * C::IN_BUF + cont_off(k.fold_tap * C::CC + l4);
                } else {
                    int G = 4 * s + l4, cg = G / C::Q, q = G - cg * C::Q;
                    if (cg >= n_full) {
                        // the short last chunk (fewer than CC cells): slot = (tap, cell) over its own n_rem cells; beyond it the
                        // padding slots of the last step (zero weights, any valid address)
                        const int q2 = G - n_full * C::Q;
                        cg = n_full < k.n_chunks ? n_full : k.n_chunks - 1;
                        q = (n_rem > 0 && q2 < C::TAPS * n_rem) ? (q2 / n_rem) * C::CC + (q2 % n_rem) : 0;
                    }
                    off = (cg & 1) * C::IN_BUF + cont_off(q);
                }
            } else {
                const int ch = s / C::NSTEP;
                off = (ch & 1) * C::IN_BUF + C::slot_lds_off(s - ch * C::NSTEP, l4);
            }
            e.bofs[l4] = (uint16_t)(off / 16);
        }
        // ---- prefetch: a share of the next chunk's tile (stages only issue at their first step)
        const int stage = s / C::SPS;
        if (s % C::SPS != 0) continue;
        int pf, r0, rstride;
        if (C::CONT) {
            const int ch = (4 * s) / C::Q;
            pf = ch + 1;
            // buffer pf & 1 is free once the last slot of chunk pf - 2 is done and must be full before the first slot of chunk
            // pf: steps [ws, we], i.e. the stages that lie wholly inside them; a stage that straddles two chunks issues nothing
            const int ws = pf >= 2 ? (C::Q * (pf - 1) - 1) / 4 + 1 : 0;
            const int we = (C::Q * pf) / 4 - 1;
            const int gs = (ws + C::SPS - 1) / C::SPS, ge = (we + 1) / C::SPS - 1;
            r0 = (stage >= gs && stage <= ge) ? stage - gs : C::NR;
            rstride = ge - gs + 1;
            if (folded && s >= n_stages_a) {        // a folded chunk lasts one step: the next one is fetched whole during it
                pf = chunks1 + (s - n_stages_a) + 1;
                r0 = 0;
                rstride = 1;
            }
        } else {
            const int ch = s / C::NSTEP;
            pf = ch + 1;
            r0 = s - ch * C::NSTEP;
            rstride = C::NSTEP;
        }
        if (pf < k.n_chunks) {
            if (r0 >= C::NR) continue;
            unsigned rmask = 0;
            for (int r = r0; r < C::NR; r += rstride) rmask |= 1u << r;
            fetch(e, pf, rmask, r0 == 0 && pf == chunks1);
        } else if (pf == k.n_chunks && !folded && !vol && !k.has_in2) {
            // the window of the last chunk: persistent workgroups fetch chunk 0 of their NEXT tile here (same schedule as any
            // other chunk, into the buffer the chunk numbering -- continued across tiles -- gives it)
            if (C::CONT) {
                const int ws = pf >= 2 ? (C::Q * (pf - 1) - 1) / 4 + 1 : 0;
                const int we = n_stages - 1;                      // (a short last chunk ends early)
                const int gs = (ws + C::SPS - 1) / C::SPS, ge = (we + 1) / C::SPS - 1;
                if (stage < gs || stage > ge) continue;
                r0 = stage - gs;
                rstride = ge - gs + 1;
            }
            if (r0 >= C::NR) continue;
            unsigned rmask = 0;
            for (int r = r0; r < C::NR; r += rstride) rmask |= 1u << r;
            fetch(e, 0, rmask, r0 == 0);
            e.dma |= SPLIT_DMA_NEXT;
            if (k.n_chunks & 1) e.dma |= SPLIT_DMA_BUF;           // chunk n_chunks of the continued numbering
            next_rounds |= rmask;
        }
    }
    // entry 0 says whether the plan holds a complete next-tile fetch (the library launches the persistent kernel only then)
    if (next_rounds == (1u << C::NR) - 1u) out[0].dma |= SPLIT_DMA_NEXT;
    // second copy for tiles that start in the other input buffer (persistent workgroups, odd number of chunks per tile)
    const size_t n = out.size();
    out.resize(2 * n);
    for (size_t i = 0; i < n; ++i) {
        SplitStep e = out[i];
        for (int l4 = 0; l4 < 4; ++l4) {
            const int off = e.bofs[l4] * 16;
            e.bofs[l4] = (uint16_t)((off >= C::IN_BUF ? off - C::IN_BUF : off + C::IN_BUF) / 16);
        }
        if (e.dma & SPLIT_DMA_ANY) e.dma ^= SPLIT_DMA_BUF;
        out[n + i] = e;
    }
}

// EPI: as conv_mfma.h (EPI_PLAIN / EPI_RES / EPI_RES_POST / EPI_HEAD) with split outputs (the head: fp32 scores);
// EPI_PLAIN_F32 = plain epilogue storing fp32 planes, for a layer whose consumer is not on the 2xf16 path.
// EPI_POOL = plain epilogue followed by MaxPool2d(2) (floor): the encoder convs of the U-Nets, whose un-pooled output
// nobody else reads (denoising/models.py:81-97) -- the 2x2 maximum is taken on the fp32 values in registers (row
// pairs inside the wave, column pairs across neighbouring lanes) and only the pooled tensor is stored.
enum { EPI_PLAIN_F32 = 5, EPI_POOL = 6 };
// ABL: timing-ablation switches for tools/split_ablate.hip only (production kernels use ABL = 0; results are not
// meaningful otherwise):  1 no epilogue loads / stores   2 no per-step DMA issue   4 no per-step barrier
//   8 fragment reads only for m = 0 (operands reused)   16 no MFMAs   32 no residual loads   64 no stores
//   128 no per-step weight DMA   256 no per-step input DMA
//   4096 input DMA without the offset table (contiguous dummy source)   8192 input DMA from a 2 MB window (always L2 hits)
//   16384 DMA issued but never waited for inside the K loop (what the latency of a stage's own prefetch costs)
//   524288 / 1048576 s_setprio 1 for the upper / lower half of an 8-wave workgroup during the K loop (experiment)
//   262144 the round-5 epilogue accesses: 8-byte stores / residual loads per lane instead of the paired 16-byte ones
//   2048 clock probe: every workgroup adds its duration in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime)
//        to the two 64-bit counters behind a.flag -> effective clock of the variant (DVFS: the chip runs at its power limit)
// MODE: what the instantiation can do besides a plain single-source 2-D conv -- bit 0: a second source (`in2`: fused upsample +
// concat, space-to-depth skip cell, folded projection), bit 1: plane-stacked 3-D.  The library launches MODE 0 whenever a layer
// needs neither (every layer of the scoring networks but the folded ones, most of the U-Nets'): its per-step scalar code then
// carries none of the other modes' selects and branches.
// (the body of the kernels below: `a` = the launch's arguments, (bid_x, bid_y, bid_z) of a (grid_x, grid_y, .) grid = this
// workgroup's place in it -- blockIdx / gridDim for a launch of its own, the entry's share of a batched launch otherwise)
template <class C, int EPI, int ABL, int MODE>
__device__ __forceinline__ void conv_split_body(const SplitArgs& a, const unsigned bid_x, const unsigned bid_y, const unsigned bid_z,
                                                const unsigned grid_x, const unsigned grid_y) {
    constexpr int D = C::D, MW = C::MW, NW = C::NW, NFC = C::NFC;
    constexpr bool HAS2 = (MODE & 1) != 0, VOLM = (MODE & 2) != 0;
    // MODE bit 3 (with bits 0 and 1): every chunk of a plane-stacked two-source launch comes from ONE tensor (SplitArgs::
    // vol_srcmajor) -- the source is a scalar choice per chunk, as in 2-D, instead of a per-lane one
    constexpr bool UNI2 = (MODE & 8) != 0;
    static_assert(!UNI2 || (HAS2 && VOLM), "MODE 11 = two sources, plane-stacked, chunk-uniform");
    // MODE bit 2: PERSISTENT workgroups.  The grid is a few workgroups per CU; each walks its share of the tiles (XCD-aware
    // order) and, during the last chunk of a tile's K loop, fetches the first chunk and the first weight stage of ITS NEXT tile
    // (the plan's SPLIT_DMA_NEXT entries): the table computation, the first fetch and its full memory latency -- 6 - 9 k cycles
    // per tile, 4 % (128-channel 3x3 tiles) to 20 % (48 channels) of a tile's time, profiles/r03_phase_breakdown.txt -- leave
    // the critical path.  Plain single-source 2-D layers without the fused head only (MODE 4).
    constexpr bool PERSIST = (MODE & 4) != 0;
    static_assert(!PERSIST || (MODE == 4 && EPI != EPI_HEAD), "persistent tiles: plain single-source layers");
    const uint4* const in2 = HAS2 ? a.in2 : nullptr;
    const int fold_cells = HAS2 ? a.fold_cells : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned* lds_tab = reinterpret_cast<unsigned*>(lds + C::OFF_TAB);

    const int tid = threadIdx.x;
    unsigned long long probe_c0 = 0, probe_r0 = 0, probe_pro = 0, probe_loop = 0;     // (+ phase breakdown: prologue / K loop / epilogue)
    // ABL 4096 (with 2048): where a step's cycles go, per wave -- DMA code | channel fragments 0 .. MW-2 | DMA drain + barrier |
    // last fragment + next step's first requests; reported for wave 0 and wave WAVES/2 (the two waves of one SIMD)
    unsigned long long seg[4] = {0, 0, 0, 0}, seg_t = 0;
    auto seg_mark = [&](int i) {
        if constexpr ((ABL & 4096) != 0) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (i >= 0) seg[i] += t - seg_t;
            seg_t = t;
        }
    };
    if constexpr ((ABL & 2048) != 0) {
        probe_c0 = __builtin_readcyclecounter();
        probe_r0 = __builtin_amdgcn_s_memrealtime();
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    // ---- tile geometry.  One tile per workgroup (XCD-aware order within a grid-z slice, see conv_mfma.h) or, PERSIST, a
    // sequence of tiles: linear tile index L = (z * tiles_y + by) * tiles_x + bx; the tiles are dealt to the XCDs in eight
    // contiguous runs (block b runs on XCD b % 8: neighbouring tiles share their halo in that XCD's L2) and within a run
    // round-robin to the XCD's workgroups.
    int y0, x0, cogz, oz, pad_x, pad_y, pad_z, oox, ooy, ooz, phase, ybase, xbase;
    auto set_tile = [&](int bx, int by, int bz) {
        y0 = a.wy0 + (by / D) * (C::TH * D) + (by % D);
        x0 = a.wx0 + bx * C::TW;
        cogz = bz % a.ncz;
        oz = bz / a.ncz;                                     // output plane of the launch lattice (0 in 2-D)
        pad_x = a.pad_x; pad_y = a.pad_y; pad_z = a.pad_z; oox = a.oox; ooy = a.ooy; ooz = a.ooz; phase = 0;
        if (a.nphase > 0) {
            phase = oz % a.nphase;
            oz /= a.nphase;
            oox = phase & 1; ooy = (phase >> 1) & 1; ooz = (phase >> 2) & 1;
            pad_x = (a.phase_k / 2 - oox + 1) / 2;
            pad_y = (a.phase_k / 2 - ooy + 1) / 2;
            pad_z = (a.phase_k / 2 - ooz + 1) / 2;
        }
        oz += a.wz0;                                         // (the launch covers the planes of its window only)
        ybase = y0 - pad_y; xbase = x0 - pad_x;
    };
    auto set_tile_linear = [&](unsigned L) {
        const unsigned txy = (unsigned)(a.tiles_x * a.tiles_y);
        const unsigned bz = L / txy, rem = L - bz * txy, by = rem / (unsigned)a.tiles_x;
        set_tile((int)(rem - by * (unsigned)a.tiles_x), (int)by, (int)bz);
    };
    unsigned tile_L = 0, tile_end = 0, tile_stride = 1;       // PERSIST: this workgroup's tiles L, L + stride, ... < end
    if constexpr (PERSIST) {
        const unsigned nt = (unsigned)a.n_tiles;
        const unsigned xcd = bid_x & 7u, j = bid_x >> 3, q = nt / 8, r = nt % 8;
        const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile_end = start + q + (xcd < r ? 1u : 0u);
        tile_stride = grid_x >> 3;
        tile_L = start + j;
        if (tile_L >= tile_end) return;                       // (more workgroups than tiles in this XCD's run)
        set_tile_linear(tile_L);
    } else {
        int bx = (int)bid_x, by = (int)bid_y;
        if (a.xcd_swizzle) {
            const unsigned nwg = grid_x * grid_y, orig = bid_x + grid_x * bid_y;
            const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
            const unsigned wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
            if (a.xcd_swizzle == 2) {
                // PATCH raster: the grid is padded to whole 8 x 4 blocks of tiles and an XCD's run of workgroup ids walks them
                // block by block, so the 32 tiles its CUs hold at a time form a compact patch whose halos overlap in that
                // XCD's L2 (a row-major run shares columns only: 25 GB fetched for 8.7 GB of tensors on the 5x5 d4 layer).
                // Rows are taken phase-major -- r = phase * bands + band -- because the D row phases of a dilated layer share
                // nothing, while neighbouring bands of one phase share K - 1 rows.
                const unsigned blk = wgid >> 5, in = wgid & 31u, nbx = grid_x >> 3;
                const unsigned byb = blk / nbx, tr = byb * 4 + (in >> 3), bands = (unsigned)a.tiles_y / (unsigned)D;
                bx = (int)((blk - byb * nbx) * 8 + (in & 7u));
                if (bx >= a.tiles_x || tr >= (unsigned)a.tiles_y) return;       // (padding of the last blocks)
                const unsigned ph = tr / bands;
                by = (int)((tr - ph * bands) * (unsigned)D + ph);
            } else {
                bx = (int)(wgid % grid_x);
                by = (int)(wgid / grid_x);
            }
        }
        set_tile(bx, by, (int)bid_z);
    }
    const bool vol = VOLM && (a.KZ > 1 || a.Din > 1);        // plane-stacked 3-D addressing

    // marks a cell outside the image / past the last channel.  16-byte aligned: as the offset of a buffer load all four of its
    // dwords lie outside every descriptor (no 32-bit wrap), and the hardware then writes ZEROS to the LDS -- checked on the
    // MI355X by tools/bufdma_probe.hip (the range check is per dword)
    constexpr unsigned OOB = 0xfffffff0u;
    // how the LDS-DMA is issued: buffer loads (descriptor per chunk and plane, 32-bit lane offsets straight from the table,
    // out-of-range lanes zero-filled by the hardware) or, ABL 65536, the round-2 global_load_lds forms
    constexpr bool BUFDMA = (ABL & 65536) == 0;
    // ---- the chunk-invariant global byte offset of every LDS cell of a source
    const bool ups = (a.H1 != a.Hin) || (a.W1 != a.Win);
    // (tid and the sizes nearest_src() divides are passed in: the co-group loop hands over opaque copies, so that nothing of this prologue is
    // hoisted out of that loop and kept in registers across the K loop)
    auto compute_offsets = [&](bool second, int tid, int H1, int W1, int Hup, int Wup, int ybase, int xbase) {
        const bool fold2 = second && fold_cells > 0;       // the folded source has its own size and origin
        const int Hs = fold2 ? a.in2_H : second ? a.Hin : H1, Ws = fold2 ? a.in2_W : second ? a.Win : W1;
        const int Hv = fold2 ? a.in2_H : a.Hin, Wv = fold2 ? a.in2_W : a.Win;
#pragma unroll 1
        for (int i = 0; i < C::NR; ++i) {
            const int g = i * C::THREADS + tid;
            if (g < C::NPC) {
                const int c = g / C::CELL_STRIDE;
                const int rem = g - c * C::CELL_STRIDE;
                const int r = rem / C::ITW, x = rem - r * C::ITW;
                const int gy = ybase + r * D + (fold2 ? a.in2_oy : 0), gx = xbase + x + (fold2 ? a.in2_ox : 0);
                unsigned off = OOB;
                if ((unsigned)gy < (unsigned)Hv && (unsigned)gx < (unsigned)Wv) {
                    int sy = gy, sx = gx;
                    if (!second && ups) { sy = nearest_src(gy, H1, Hup); sx = nearest_src(gx, W1, Wup); }
                    // 3-D: the in-plane part only; the (cell, plane) part is added per chunk in fetch
                    off = vol ? (unsigned)(((size_t)sy * Ws + sx) * 16) : (unsigned)((((size_t)c * Hs + sy) * Ws + sx) * 16);
                }
                lds_tab[g] = off;              // read back by the same thread only
            }
        }
    };
    const size_t plane1 = (size_t)a.cells_in1 * a.H1 * a.W1 * (vol ? a.Din : 1);   // cells per plane of `in`
    const size_t hw2 = fold_cells > 0 ? (size_t)a.in2_H * a.in2_W : (size_t)a.Hin * a.Win;      // one cell plane of `in2`
    const size_t plane2 = (size_t)(a.cells_in - a.cells_in1) * hw2 * (vol ? a.Din : 1);           // ... of `in2`
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)lds;
    const void* zsrc = uniform_ptr(a.zeros);
    // the plan is read through the scalar cache: one s_load_dwordx8 per step, a step ahead of its use
    typedef uint32_t StepW __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) StepW* plan_ptr_t;
    const plan_ptr_t plan0 = (plan_ptr_t)a.plan;

    // One LDS-DMA instruction moves 16 bytes per lane, 1 KiB per wave.  Two ways to issue it:
    //  * per-lane 64-bit source address through the builtin (glds16): lanes whose cell lies outside the image (or past the last
    //    channel) point at the zero block, a lane of the plane-stacked 3-D mode at its own source tensor -- one instruction per
    //    piece, no divergent paths, M0 set by the compiler.  Fewer instructions: what the narrow tiles, bound by the issue of
    //    everything that is not an MFMA, want (-2 ... -5 %).
    //  * wave-uniform base + 32-bit lane offset in inline asm (glds_b128), with separate paths for waves that hold out-of-image
    //    lanes: no 64-bit address arithmetic in the vector unit.  The 128-channel 8-wave tiles run at 256 VGPRs and lose 1.5 %
    //    with the per-lane form (same-box A/B of the fused-head kernel), so they keep this one (SplitCfg::UNIFORM_DMA).
    auto glds16 = [&](const void* src, unsigned lds_addr) {
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                         (__attribute__((address_space(3))) void*)(size_t)lds_addr, 16, 0, 0);
    };
    //  * (round 3, the default) buffer loads: a descriptor per (chunk, plane) built from scalars, the table entry as the 32-bit
    //    lane offset, cells outside the image carry the offset OOB and come back as zeros -- no zero block, no per-lane
    //    pointers, no divergent paths: table read + 2 instructions per round.
    auto make_srd = [&](const void* base, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_ptr(base)), 0,
                                                 (int)__builtin_amdgcn_readfirstlane((unsigned)bytes), 0x00020000);
    };
    auto bdma16 = [&](__amdgpu_buffer_rsrc_t srd, unsigned voff, unsigned soff, unsigned lds_addr) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(size_t)lds_addr, 16, voff, soff, 0, 0);
    };
    // The rounds `rounds` (bit r) of the input tile of the chunk that plan entry P describes -> its input buffer (both planes);
    // (tid, wave): whose share.  Everything but the table lookup and the DMA itself is scalar.
    auto fetch = [&](const StepW& P, unsigned rounds, int tid, int wave) {
        const unsigned d = P[2];
        const int buf = (d & SPLIT_DMA_BUF) ? 1 : 0;
        const bool second = HAS2 && (d & SPLIT_DMA_SRC2) != 0;
        const unsigned ncell = (d >> SPLIT_DMA_NCELL_SHIFT) & 7u;       // 0: every cell of the chunk exists
        const uint4* chunk = second ? in2 + (size_t)P[3] * (C::CC * hw2) : a.in + (size_t)P[3] * (C::CC * (size_t)a.H1 * a.W1);
        const unsigned char* bhi = reinterpret_cast<const unsigned char*>(uniform_ptr(vol ? ((UNI2 && second) ? in2 : a.in) : chunk));
        const size_t lo_delta0 = ((second && (!vol || UNI2)) ? plane2 : plane1) * 16;       // bytes from a hi cell to its lo cell
        // buffer descriptors: the chunk's CC cell planes (2-D) or one half of the whole tensor (plane-stacked 3-D), hi and lo
        const size_t ext = vol ? ((UNI2 && second) ? plane2 : plane1) * 16 : (second ? (size_t)C::CC * hw2 : (size_t)C::CC * a.H1 * a.W1) * 16;
        const __amdgpu_buffer_rsrc_t srd_hi = make_srd(bhi, ext), srd_lo = make_srd(bhi + lo_delta0, ext);
        // plane-stacked 3-D: virtual cell v = kz * cells + c of the chunk is cell c of input plane oz + kz - pad_z, of `in` or
        // `in2` (32-bit byte offsets from the tensor start: the host keeps plane-stacked tensors below 4 GiB per half)
        constexpr int CCV = C::CC < 4 ? C::CC : 4;
        unsigned coff[CCV], cflag[CCV];
        if (vol) {
#pragma unroll
            for (int j = 0; j < CCV; ++j) {
                const unsigned cv = P[4 + j];
                const int iz = oz + (int)((cv >> 2) & 63u) * D - pad_z;       // (dilated 3-D convs: planes D apart, like the rows)
                const bool ok = (cv & 1u) && (unsigned)iz < (unsigned)a.Din;
                coff[j] = (unsigned)((((size_t)(cv >> 8) * a.Din + iz) * a.Hin) * a.Win * 16);
                cflag[j] = (ok ? 1u : 0u) | ((HAS2 && !UNI2) ? (cv & 2u) : 0u);
            }
        }
#pragma unroll
        for (int r = 0; r < C::NR; ++r) {
            if (!((rounds >> r) & 1u)) continue;
            const int g = r * C::THREADS + tid;
            if ((r + 1) * C::THREADS <= C::NPC || g < C::NPC) {
                unsigned off = lds_tab[g];
                size_t lo_delta = lo_delta0;
                bool lane2 = false;                  // 3-D: this lane's cell comes from `in2`
                if (!vol) {
                    if (ncell != 0 && g >= (int)ncell * C::CELL_STRIDE) off = OOB;      // cells past the last channel
                } else {
                    unsigned co = coff[0], cf = cflag[0];
#pragma unroll
                    for (int j = 1; j < CCV; ++j)
                        if (g >= j * C::CELL_STRIDE) { co = coff[j]; cf = cflag[j]; }
                    if (!(cf & 1u)) off = OOB;
                    else if (off != OOB) { off += co; lane2 = HAS2 && !UNI2 && (cf & 2u) != 0; }
                }
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * C::IN_BUF + (r * C::THREADS + wave * 64) * 16));
                if constexpr (BUFDMA) {
                    if (VOLM && HAS2 && !UNI2 && __any(lane2)) {
                        // (plane-stacked 3-D with a second source: the lanes of a wave may straddle cells of both tensors)
                        if (lane2) {
                            const __amdgpu_buffer_rsrc_t s2h = make_srd(in2, plane2 * 16), s2l = make_srd(in2 + plane2, plane2 * 16);
                            bdma16(s2h, off, 0, dst);
                            bdma16(s2l, off, 0, dst + C::PLANE_BYTES);
                        } else {
                            bdma16(srd_hi, off, 0, dst);
                            bdma16(srd_lo, off, 0, dst + C::PLANE_BYTES);
                        }
                    } else {
                        bdma16(srd_hi, off, 0, dst);
                        bdma16(srd_lo, off, 0, dst + C::PLANE_BYTES);
                    }
                } else if constexpr (C::UNIFORM_DMA) {
                    const void* blo = bhi + lo_delta;
                    if (!__any(off == OOB || lane2)) {
                        glds_b128(off, bhi, dst);
                        glds_b128(off, blo, dst + C::PLANE_BYTES);
                    } else {
                        if (off == OOB) {
                            glds_b128(0u, zsrc, dst);
                            glds_b128(0u, zsrc, dst + C::PLANE_BYTES);
                        } else if (lane2) {
                            const void* b2 = uniform_ptr(in2);
                            const void* b2lo = uniform_ptr(in2 + plane2);
                            glds_b128(off, b2, dst);
                            glds_b128(off, b2lo, dst + C::PLANE_BYTES);
                        } else {
                            glds_b128(off, bhi, dst);
                            glds_b128(off, blo, dst + C::PLANE_BYTES);
                        }
                    }
                } else {
                    const unsigned char* src = bhi;
                    if (lane2) { src = reinterpret_cast<const unsigned char*>(in2); lo_delta = plane2 * 16; }
                    const bool oob = off == OOB;
                    const unsigned char* shi = oob ? reinterpret_cast<const unsigned char*>(zsrc) : src + off;
                    const unsigned char* slo = oob ? reinterpret_cast<const unsigned char*>(zsrc) : src + off + lo_delta;
                    glds16(shi, dst);
                    glds16(slo, dst + C::PLANE_BYTES);
                }
            }
        }
    };
    // weights of stage `stg` (its steps that exist: `bytes` = min(SPS, steps left) * W_STEP_BYTES) -> weight buffer `buf`
    auto issue_weights = [&](const unsigned char* wcog, int stg, int bytes, int buf, int tid, int wave) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(uniform_ptr(wcog + (size_t)stg * C::W_STAGE_BYTES));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(C::OFF_W + buf * C::W_STAGE_BYTES + wave * 1024));
        if constexpr (BUFDMA) {
            const __amdgpu_buffer_rsrc_t srd = make_srd(base, (size_t)bytes);
#pragma unroll
            for (int i = 0; i < C::WR; ++i)
                if ((i * C::WAVES + wave) * 1024 < bytes) bdma16(srd, (unsigned)tid * 16u, (unsigned)(i * C::THREADS * 16), dst + i * C::THREADS * 16);
            return;
        }
#pragma unroll
        for (int i = 0; i < C::WR; ++i)
            if ((i * C::WAVES + wave) * 1024 < bytes) {                // whole waves (1 KiB each)
                if constexpr (C::UNIFORM_DMA) glds_b128((unsigned)(i * C::THREADS + tid) * 16u, base, dst + i * C::THREADS * 16);
                else glds16(base + (unsigned)(i * C::THREADS + tid) * 16u, dst + i * C::THREADS * 16);
            }
    };

    // per-lane LDS read bases (bytes)
    const unsigned b_lane = (unsigned)(((wave * C::RPW) * C::ITW + l15) * 16);
    const unsigned a_lane = (unsigned)(C::OFF_W + lane * 16);
    // LDS address of this lane's B fragments of a step: its lane group's entry of the step's plan (buffer + tap + cell)
    auto b_frag_base = [&](const StepW& P, int l4) -> const unsigned char* {
        const unsigned w = (l4 & 2) ? P[1] : P[0];
        return lds + b_lane + (((w >> ((l4 & 1) * 16)) & 0xffffu) << 4);
    };

    float hsum[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) hsum[n] = 0.f;
    bool big = false;

    const bool folded = C::CONT && C::SPS == 1 && fold_cells > 0;
    const int vcells = (folded ? a.cells_in1 : a.cells_in) * (vol ? a.KZ : 1);        // (virtual) cells of the K loop (all taps)
    const int n_stages_a = C::CONT ? C::cont_stages(vcells) : a.n_chunks * C::NSTEP;
    const int n_stages = n_stages_a + (folded ? fold_cells / C::CC : 0);    // + one step per folded chunk
    const size_t w_cog_bytes = (size_t)n_stages * C::W_STEP_BYTES;

    // PERSIST: parities of the input / weight buffers at the start of the current tile (chunks and stages are numbered on across
    // tiles: a tile with an odd number of them leaves the next one starting in the other buffer; the plan has a copy for either
    // input parity), and whether the current tile's first chunk + first weight stage were already fetched by its predecessor
    const int n_wstages = (n_stages + C::SPS - 1) / C::SPS;
    int par_in = 0, par_w = 0;
    bool prefetched = false;
    auto wcog_of = [&](int phase, int cog) {
        return reinterpret_cast<const unsigned char*>(a.wpk) + (size_t)phase * a.w_phase_bytes + (size_t)cog * w_cog_bytes;
    };
    for (;;) {                                  // tiles of this workgroup (exactly one when !PERSIST)
    const bool has_next = PERSIST && tile_L + tile_stride < tile_end;
    int n_ybase = 0, n_xbase = 0, n_phase = 0, n_cog = 0;     // origin, phase and co-group of the next tile
    if (PERSIST && has_next) {
        const int sy0 = y0, sx0 = x0, scogz = cogz, soz = oz, spx = pad_x, spy = pad_y, spz = pad_z, sox = oox, soy = ooy,
                  soz2 = ooz, sph = phase, syb = ybase, sxb = xbase;
        set_tile_linear(tile_L + tile_stride);
        n_ybase = ybase; n_xbase = xbase; n_phase = phase; n_cog = cogz;
        y0 = sy0; x0 = sx0; cogz = scogz; oz = soz; pad_x = spx; pad_y = spy; pad_z = spz; oox = sox; ooy = soy; ooz = soz2;
        phase = sph; ybase = syb; xbase = sxb;
    }
    const plan_ptr_t plan = plan0 + (PERSIST && par_in ? n_stages + 1 : 0);
    for (int cg = 0; cg < a.cog_inner; ++cg) {
        const int cog = cogz * a.cog_inner + cg;
        const unsigned char* wcog = wcog_of(phase, cog);
        const float* wscale = a.wscale + (size_t)phase * a.ws_phase_stride;
        f32x4 acc[MW][NW];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int n = 0; n < NW; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

        unsigned long long probe_t0 = 0;
        if constexpr ((ABL & 2048) != 0) probe_t0 = (cg == 0 && !prefetched) ? probe_c0 : __builtin_readcyclecounter();
        __syncthreads();                       // previous co-group done with the buffers
        // the prologue of a co-group works from opaque copies of the thread id and the source size: its lane-dependent
        // values are then computed here and die here, instead of being hoisted out of this loop and spilled across the
        // K loop of the fused-head variant (which runs two co-groups per tile at 256 VGPRs)
        int tid_p = tid, H1_p = a.H1, W1_p = a.W1, Hup_p = a.Hin, Wup_p = a.Win;
        asm volatile("" : "+v"(tid_p), "+s"(H1_p), "+s"(W1_p), "+s"(Hup_p), "+s"(Wup_p));
        const int wave_p = __builtin_amdgcn_readfirstlane(tid_p >> 6), l4_p = (tid_p & 63) >> 4;
        StepW P = plan[1];                     // step 0
        if (!(PERSIST && prefetched)) {
            const StepW P0 = plan0[0];         // the fetch of chunk 0
            compute_offsets((P0[2] & SPLIT_DMA_SRC2) != 0, tid_p, H1_p, W1_p, Hup_p, Wup_p, ybase, xbase);
            fetch(P0, (1u << C::NR) - 1u, tid_p, wave_p);
            issue_weights(wcog, 0, (n_stages < C::SPS ? n_stages : C::SPS) * C::W_STEP_BYTES, 0, tid_p, wave_p);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // The first fragments of a step -- all its B fragments (hi, lo) and A(0) -- are loop-carried registers: they are
        // refreshed IN PLACE for step s + 1 during the last channel fragment of step s, each right after the last MFMA
        // that reads its register has been issued (an MFMA reads its A/B operands at issue; the LDS data lands >= 64
        // cycles later), so the pipelining across the barrier costs no extra registers and no copies.
        f16x8 bh[NW], bo[NW], ah[2], ao[2];
        auto b_off = [&](int n) { return ((n / NFC) * C::ITW + (n % NFC) * 16) * 16; };
        {
            const unsigned char* bl = b_frag_base(P, l4_p);
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                bh[n] = *reinterpret_cast<const f16x8*>(bl + b_off(n));
                bo[n] = *reinterpret_cast<const f16x8*>(bl + b_off(n) + C::PLANE_BYTES);
            }
            const unsigned a_lane_p = (unsigned)(C::OFF_W + (tid_p & 63) * 16 + (PERSIST ? par_w * C::W_STAGE_BYTES : 0));
            ah[0] = *reinterpret_cast<const f16x8*>(lds + a_lane_p);
            ao[0] = *reinterpret_cast<const f16x8*>(lds + a_lane_p + MW * 1024);
        }

        unsigned long long probe_t1 = 0;
        if constexpr ((ABL & 2048) != 0) probe_t1 = __builtin_readcyclecounter();
        // ABL 524288 / 1048576 (experiment): static issue priority for the second-dispatched half of an 8-wave workgroup -- the
        // arbitration loser of every segment (MI355X_MICROARCH.md, two waves per SIMD, item 4) -- / for the first half
        if constexpr ((ABL & 524288) != 0) { if (C::WAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1); }
        if constexpr ((ABL & 1048576) != 0) { if (C::WAVES == 8 && wave < 4) __builtin_amdgcn_s_setprio(1); }
#pragma unroll 1
        for (int s = 0; s < n_stages; ++s) {
            const int stage = s / C::SPS, sub = s - stage * C::SPS;      // (SPS = 1: stage = s, sub = 0)
            const bool stage_end = sub == C::SPS - 1 || s == n_stages - 1;
            // the next step's plan entry: requested now, first used behind this step's barrier
            const StepW Pn = plan[1 + (s + 1 < n_stages ? s + 1 : s)];
            // ---- prefetch by DMA: the weights of the next stage, a share of the next chunk's input tile
            auto step_dma = [&]() {
            if (!(ABL & 2) && sub == 0) {
                // issuer_half: the upper four waves issue every piece of the step (their SIMD partners, waves w - 4,
                // start their MFMAs at once and keep the matrix core busy meanwhile)
                // (not with a second source whose offset table is rewritten mid-loop by each thread for itself -- except the
                // folded projection, which pays one extra barrier at the switch instead)
                const bool iss = C::ISSUER_HALF && a.issuer_half && (!in2 || folded);
                const int reps = iss ? (wave >= 4 ? 2 : 0) : 1;
                if constexpr (!(ABL & 128)) {
                    for (int rep = 0; rep < reps; ++rep) {
                        const int vw = iss ? wave - 4 * (1 - rep) : wave, vt = iss ? tid - 256 * (1 - rep) : tid;
                        const int left = n_stages - (stage + 1) * C::SPS;            // steps of the next stage
                        const int wb = PERSIST ? (stage + 1 + par_w) & 1 : (stage + 1) & 1;
                        if (left > 0) issue_weights(wcog, stage + 1, (left < C::SPS ? left : C::SPS) * C::W_STEP_BYTES, wb, vt, vw);
                        else if (PERSIST && has_next)      // the last stage: the first stage of the next tile takes its place
                            issue_weights(wcog_of(n_phase, n_cog), 0, (n_stages < C::SPS ? n_stages : C::SPS) * C::W_STEP_BYTES, wb, vt, vw);
                    }
                }
                const bool next_fetch = (P[2] & SPLIT_DMA_NEXT) != 0;      // chunk 0 of the NEXT tile (persistent workgroups only)
                if (!(ABL & 256) && (P[2] & SPLIT_DMA_ANY) && (!next_fetch || (PERSIST && has_next))) {
                    if (HAS2 && (P[2] & SPLIT_DMA_SWITCH)) {
                        compute_offsets(true, tid, a.H1, a.W1, a.Hin, a.Win, ybase, xbase);         // switching to the second source
                        if (iss) __syncthreads();                                     // the issuing waves read other threads' entries
                    }
                    if (PERSIST && next_fetch && (P[2] & SPLIT_DMA_SWITCH)) {
                        // the current tile's last fetch was issued in an earlier step: the table now becomes the next tile's
                        compute_offsets(false, tid, a.H1, a.W1, a.Hin, a.Win, n_ybase, n_xbase);
                        if (iss) __syncthreads();
                    }
                    for (int rep = 0; rep < reps; ++rep) {
                        const int vw = iss ? wave - 4 * (1 - rep) : wave, vt = iss ? tid - 256 * (1 - rep) : tid;
                        fetch(P, P[2] & SPLIT_DMA_ROUNDS, vt, vw);
                    }
                }
            }
            };
            // (issued at the top of the step.  Issuing it between the MFMAs of the middle channel fragments instead was measured:
            // +-0 on the narrow tiles, -8 % with issuer_half -- profiles/r03_kloop_experiments.txt)
            seg_mark(s == 0 ? -1 : 3);
            step_dma();
            seg_mark(0);
            // Fragment 0 issues its first NW MFMAs BEFORE the request for A(1): the wait in front of the step's first MFMA is an
            // lgkmcnt(0) whatever is in flight (the plan's scalar load returns out of order), so the A(1) request would be
            // waited for there as well (+1.5 % on the fused-head kernel; ABL 131072 = the old order)
            constexpr bool MFMA_FIRST = (ABL & 131072) == 0;
            // ---- the step's MFMAs, register-pipelined across the step barrier:
            //   channel fragments 0 .. MW-2 (A(m + 1) requested one fragment ahead), then the DMA drain + barrier, then the
            //   MFMAs of the last channel fragment -- which need registers only -- interleaved with the requests for the
            //   B fragments and A(0) of step s + 1.  The matrix core is not left idle for an LDS round trip after every
            //   barrier (150 - 250 cycles of a step of 768 (MT = 64) ... 3072 (MT = 128, 8 waves) cycles).
            //   MFMA order within a channel fragment: ah*bo, ah*bh, ao*bh -- the lo halves of B are released first.
            const int wpar = PERSIST ? par_w : 0;
            const unsigned char* al = lds + a_lane + ((stage + wpar) & 1) * C::W_STAGE_BYTES + sub * C::W_STEP_BYTES;
            const int stage_n = (s + 1) / C::SPS;
            const unsigned char* al_next = lds + a_lane + ((stage_n + wpar) & 1) * C::W_STAGE_BYTES + (s + 1 - stage_n * C::SPS) * C::W_STEP_BYTES;
            const unsigned char* bl_next = nullptr;
            constexpr bool A0_EARLY = (MW % 2 == 0);          // slot 0 of ah / ao is free during the last fragment (slot 1)
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int t = m & 1;
                if (m + 1 < MW) {
                    if constexpr (!(ABL & 8)) {
                        ah[t ^ 1] = *reinterpret_cast<const f16x8*>(al + (m + 1) * 1024);
                        ao[t ^ 1] = *reinterpret_cast<const f16x8*>(al + (MW + m + 1) * 1024);
                    } else {
                        ah[t ^ 1] = ah[t];
                        ao[t ^ 1] = ao[t];
                    }
                } else {
                    // ---- last channel fragment.  First pin the order of everything above (the machine scheduler otherwise
                    // sinks every fragment read to just before its first use and waits lgkmcnt(0) on it: 2 * MW exposed
                    // LDS latencies per step)
                    if constexpr (!(ABL & 16) && !(ABL & 8)) {
#pragma unroll
                        for (int mm = 0; mm + 1 < MW; ++mm) {
                            if (mm == 0 && MFMA_FIRST) {
                                __builtin_amdgcn_sched_group_barrier(0x008, NW, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NW, 0);
                            } else {
                                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // A(mm + 1)
                                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NW, 0);    // MFMAs of mm
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    seg_mark(1);
                    if (C::SPS == 1 || stage_end) {
                        if constexpr (!(ABL & 16384))
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every DMA piece this wave issued in this stage has landed
                        if constexpr (!(ABL & 4)) __syncthreads();
                    }
                    seg_mark(2);
                    bl_next = b_frag_base(Pn, l4);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bool last = m + 1 == MW;
                f16x8 a_hi = ah[t], a_lo = ao[t];
                if (last && A0_EARLY) {
                    ah[0] = *reinterpret_cast<const f16x8*>(al_next);
                    ao[0] = *reinterpret_cast<const f16x8*>(al_next + MW * 1024);
                }
                if constexpr (!(ABL & 16)) {
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bo[n], acc[m][n], 0, 0, 0);
                    if (last) {
#pragma unroll
                        for (int n = 0; n < NW; ++n) bo[n] = *reinterpret_cast<const f16x8*>(bl_next + b_off(n) + C::PLANE_BYTES);
                    }
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NW; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bh[n], acc[m][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int n = 0; n < NW; ++n) acc[m][n][0] += (float)a_hi[0] * (float)bh[n][0] + (float)a_lo[1] * (float)bo[n][1];
                    if (last) {
#pragma unroll
                        for (int n = 0; n < NW; ++n) bo[n] = *reinterpret_cast<const f16x8*>(bl_next + b_off(n) + C::PLANE_BYTES);
                    }
                }
                if (last) {
#pragma unroll
                    for (int n = 0; n < NW; ++n) bh[n] = *reinterpret_cast<const f16x8*>(bl_next + b_off(n));
                    if (!A0_EARLY) {
                        ah[0] = *reinterpret_cast<const f16x8*>(al_next);
                        ao[0] = *reinterpret_cast<const f16x8*>(al_next + MW * 1024);
                    }
                    if constexpr (!(ABL & 16) && !(ABL & 8)) {
                        if (A0_EARLY) __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
                        __builtin_amdgcn_sched_group_barrier(0x008, NW, 1);
                        __builtin_amdgcn_sched_group_barrier(0x100, NW, 1);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NW, 1);
                        __builtin_amdgcn_sched_group_barrier(0x100, A0_EARLY ? NW : NW + 2, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            P = Pn;
        }
        if constexpr ((ABL & 2048) != 0) {
            probe_loop += __builtin_readcyclecounter() - probe_t1;
            probe_pro += probe_t1 - probe_t0;
        }

        if constexpr ((ABL & (524288 | 1048576)) != 0) __builtin_amdgcn_s_setprio(0);
        // ---- epilogue: un-scale, bias, residual, eval-BN affine, activation, (fused head), split store.
        // Channel-fragment (m) outer: the per-channel constants are fetched once per m, the residual cells of
        // all NW pixel fragments are requested back to back before the first is used (the memory latency is
        // paid once per m, not once per fragment), and invalid pixels are handled by clamped addresses +
        // one predicate on the store.
        // lane coordinates re-derived behind an opaque copy of the thread index: they (and everything computed from
        // them) would otherwise stay live across the K loop next to the accumulators, and the head variant spilled them
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int l15 = tid_e & 15, l4 = (tid_e >> 4) & 3;
        const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
        const int fz = oz * a.os + ooz;                      // output plane in the full tensor (0 in 2-D)
        const size_t cplane_out = (size_t)a.Dfull * a.Hfull * a.Wfull, cplane_res = (size_t)a.Dres * a.Hres * a.Wres;   // one cell plane
        const size_t plane_out = (size_t)a.cells_out * cplane_out;
        const size_t plane_res = (size_t)a.cells_out * cplane_res;
        const size_t zoff_out = (size_t)fz * a.Hfull * a.Wfull;
        const size_t zoff_res = (size_t)(fz + (a.Dres > 1 ? a.res_crop : 0)) * a.Hres * a.Wres;     // (3-D residuals are cropped in z too)
        const bool has_bias = a.bias != nullptr;
        // Addresses of the split cells (stores, residual loads): buffer accesses.  Per fragment m one descriptor per plane (hi / lo),
        // built from scalars, covers the fragment's two cells from this tile's output plane on (num_records counts only the cells
        // that exist: the second cell of an odd cell count, padding fragments fall out of range); per pixel fragment n one 32-bit
        // byte offset per lane = pixel + (second cell of the fragment) + (second half of the cell), OOB for the pixels outside
        // the launch window -- no 64-bit lane arithmetic, no predicate: the hardware drops what is out of range (stores) or
        // returns zeros (loads).  (round 2: 2 x v_lshl_add_u64 + exec masking per access; the epilogue is issue-bound.)
        unsigned opix[NW], rpix[NW], ovo[NW], rvo[NW];
        bool okv[NW];
        if constexpr (EPI != EPI_HEAD) {       // (the fused head stores nothing here: every pixel just accumulates)
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                const int oy = y0 + (wave * C::RPW + n / NFC) * D;
                const int ox = x0 + (n % NFC) * 16 + l15;
                okv[n] = oy < a.wy1 && ox < a.wx1;               // (wy1 <= Hout, wx1 <= Wout)
                if constexpr ((ABL & 1) != 0 || (ABL & 64) != 0) okv[n] = okv[n] && (a.slope == 12345.f);
                const int cy = oy < a.Hout ? oy : a.Hout - 1, cx = ox < a.Wout ? ox : a.Wout - 1;
                const int fy = cy * a.os + ooy, fx = cx * a.os + oox;
                if constexpr (EPI == EPI_POOL) opix[n] = (unsigned)((cy >> 1) * a.Wfull + (cx >> 1));
                else opix[n] = (unsigned)(fy * a.Wfull + fx);
                rpix[n] = (unsigned)((fy + a.res_crop) * a.Wres + fx + a.res_crop);
                if constexpr (EPI == EPI_POOL) {
                    // the even row / even column of each 2x2 window stores its maximum (floor: a window must be whole)
                    okv[n] = (l15 & 1) == 0 && oy < a.wy1 && ox < a.wx1 && oy + 1 < a.Hout && ox + 1 < a.Wout;
                }
                const unsigned lane_o = (unsigned)(l4 >> 1) * (unsigned)(cplane_out * 16) + (unsigned)(l4 & 1) * 8u;
                const unsigned lane_r = (unsigned)(l4 >> 1) * (unsigned)(cplane_res * 16) + (unsigned)(l4 & 1) * 8u;
                ovo[n] = okv[n] ? opix[n] * 16u + lane_o : OOB;
                rvo[n] = rpix[n] * 16u + lane_r;              // (clamped pixel: in range whenever its cell exists)
            }
        }
        const unsigned cp16_o = (unsigned)(cplane_out * 16), cp16_r = (unsigned)(cplane_res * 16);       // bytes of one cell plane
        // WIDE accesses (round 6): an MFMA accumulator fragment leaves a lane with HALF a cell (4 of its 8 channels) of one
        // pixel, so the stores above are 8 bytes per lane and a wave's epilogue is 2 * MW * NW store instructions (and as many
        // residual loads) -- the epilogue is bound by their issue, not by bytes (cdna_hip_programming.md T21).  Pixel fragments
        // are therefore handled in PAIRS (n0, n1): one v_permlane16_swap per dword exchanges the odd 16-lane rows of n0's
        // register with the even rows of n1's, after which the lanes of an even row (l4 = 0, 2) hold BOTH halves of their cell of
        // pixel n0 and the lanes of an odd row both halves of the same cell of pixel n1: one 16-byte store per lane and plane,
        // half the instructions, every byte where it went before.  The residual cells come in the same way (16-byte loads,
        // un-swapped by the same instruction: the swap is an involution).
        constexpr bool WIDE = (NW % 2 == 0) && EPI != EPI_HEAD && EPI != EPI_PLAIN_F32 && EPI != EPI_POOL && (ABL & 262144) == 0;
        constexpr int NP = WIDE ? NW / 2 : 1;
        unsigned ovw[NP], rvw[NP];
        if constexpr (WIDE) {
            // (computed from the fragment's coordinates, not selected from opix[] / rpix[]: a lane-dependent choice between two
            // array elements is compiled into an indexed scratch access)
            const int odd = l4 & 1;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int fr = odd ? (2 * q + 1) / NFC : (2 * q) / NFC, fc = odd ? (2 * q + 1) % NFC : (2 * q) % NFC;
                const int oy = y0 + (wave * C::RPW + fr) * D;
                const int ox = x0 + fc * 16 + l15;
                bool ok = oy < a.wy1 && ox < a.wx1;
                if constexpr ((ABL & 1) != 0 || (ABL & 64) != 0) ok = ok && (a.slope == 12345.f);
                const int cy = oy < a.Hout ? oy : a.Hout - 1, cx = ox < a.Wout ? ox : a.Wout - 1;
                const int fy = cy * a.os + ooy, fx = cx * a.os + oox;
                const unsigned po = (unsigned)(fy * a.Wfull + fx), pr = (unsigned)((fy + a.res_crop) * a.Wres + fx + a.res_crop);
                ovw[q] = ok ? po * 16u + (unsigned)(l4 >> 1) * (unsigned)(cplane_out * 16) : OOB;
                rvw[q] = pr * 16u + (unsigned)(l4 >> 1) * (unsigned)(cplane_res * 16);
            }
        }
        const unsigned zo16 = (unsigned)(zoff_out * 16), zr16 = (unsigned)(zoff_res * 16);                // ... to this tile's plane
        const size_t pl16_o = plane_out * 16, pl16_r = plane_res * 16;                                    // hi -> lo
        const unsigned char* const out8 = reinterpret_cast<const unsigned char*>(a.out);
        const unsigned char* const res8 = reinterpret_cast<const unsigned char*>(a.res);
        const float slope = a.slope;
        u16x2 bigacc = {0, 0};                 // running maximum of |hi| bit patterns: >= 0x7c00 <=> an inf / NaN half
        // ... of the pixels that are stored only: a tile may overhang the launch window, and what lies outside a producer's
        // window was never written (any bit pattern)
        unsigned okmask[NW];
#pragma unroll
        for (int n = 0; n < NW; ++n) okmask[n] = (EPI != EPI_HEAD && okv[n]) ? 0x7fff7fffu : 0u;
        // where channel fragment m lies (all wave-uniform): its first channel cb within the parity (sy, sx) of a sub-pixel
        // layer, its cells cell0 (lanes l4 = 0, 1) and cell0 + 1 (l4 = 2, 3), how many of the two exist
        struct Frag { int cb, sy, sx, cell0, ncell; bool ok; };
        auto frag_of = [&](int m) {
            Frag f;
            const int covb = cog * C::MT + m * 16;
            f.cb = covb; f.sy = 0; f.sx = 0; f.ok = true;
            if (a.subpix_cout > 0) {                               // Cout % 16 == 0: one parity per fragment
                const int par = covb / a.subpix_cout;
                f.cb = covb - par * a.subpix_cout;
                f.sx = par & 1; f.sy = (par >> 1) & 1;
                f.ok = par <= 3;                                   // (padding fragments of the last co-group)
            }
            f.cell0 = f.cb >> 3;
            const int nc = f.ok ? a.cells_out - f.cell0 : 0;
            f.ncell = nc < 0 ? 0 : nc > 2 ? 2 : nc;
            return f;
        };
        // the residual cells of fragment m (all NW pixel fragments, hi and lo) are requested one fragment AHEAD of their use:
        // the latency of these loads, not the arithmetic, bounded the residual epilogues.  (The compiler cannot hoist them
        // itself: res and out may be the same tensor -- the in-place skip launch of a per-parity layer -- where every lane
        // reads exactly the cells it writes later, so running ahead of the stores of OTHER fragments is safe.)
        constexpr bool RESID = (EPI == EPI_RES || EPI == EPI_RES_POST) && !(ABL & 1) && !(ABL & 32);
        // (WIDE: entries 2q / 2q + 1 of rh hold dwords 0-1 / 2-3 of pair q's 16-byte cell until use_res un-swaps them in place)
        u32x2 rhb[2][NW], rlb[2][NW];
        auto load_res = [&](int m, u32x2 (&rh)[NW], u32x2 (&rl)[NW]) {
            const Frag f = frag_of(m);
            const unsigned ex16 = zr16 + (unsigned)(f.sy * a.Wres + f.sx) * 16u;
            const unsigned nrec = f.ncell > 0 ? (unsigned)f.ncell * cp16_r - ex16 : 0u;
            const unsigned char* rb8 = res8 + ((size_t)((unsigned)f.cell0 * (unsigned long long)cp16_r) + ex16);
            const __amdgpu_buffer_rsrc_t srd_rh = make_srd(rb8, nrec);
            const __amdgpu_buffer_rsrc_t srd_rl = make_srd(rb8 + pl16_r, nrec);
            if constexpr (WIDE) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const u32x4 ch = __builtin_amdgcn_raw_buffer_load_b128(srd_rh, (int)rvw[q], 0, 0);
                    const u32x4 cl = __builtin_amdgcn_raw_buffer_load_b128(srd_rl, (int)rvw[q], 0, 0);
                    rh[2 * q] = (u32x2){ch[0], ch[1]}; rh[2 * q + 1] = (u32x2){ch[2], ch[3]};
                    rl[2 * q] = (u32x2){cl[0], cl[1]}; rl[2 * q + 1] = (u32x2){cl[2], cl[3]};
                }
            } else {
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    rh[n] = __builtin_amdgcn_raw_buffer_load_b64(srd_rh, (int)rvo[n], 0, 0);
                    rl[n] = __builtin_amdgcn_raw_buffer_load_b64(srd_rl, (int)rvo[n], 0, 0);
                }
            }
        };
        // the loaded cells of pair q -> the half cells of fragments 2q and 2q + 1 in accumulator layout
        auto unswap_res = [&](u32x2 (&r)[NW], int q) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(r[2 * q].x, r[2 * q + 1].x, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(r[2 * q].y, r[2 * q + 1].y, false, false);
            r[2 * q] = (u32x2){s0[0], s1[0]};
            r[2 * q + 1] = (u32x2){s0[1], s1[1]};
        };
        if constexpr (RESID) load_res(0, rhb[0], rlb[0]);
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const Frag f = frag_of(m);
            const int cov0 = cog * C::MT + m * 16 + l4 * 4;       // this lane's 4 consecutive (virtual) channels: half a cell
            const int sy = f.sy, sx = f.sx, cell0 = f.cell0, ncell = f.ncell;
            const int co0 = f.ok ? f.cb + l4 * 4 : a.Cout;
            if constexpr (RESID) { if (m + 1 < MW) load_res(m + 1, rhb[(m + 1) & 1], rlb[(m + 1) & 1]); }
            u32x2 (&rh)[NW] = rhb[m & 1];
            u32x2 (&rl)[NW] = rlb[m & 1];
            // per-channel constants: one float4 each.  Every array is zero-padded to whole tiles, so the channels that pad
            // the last cell come out as 0 * acc + 0 = 0 without a mask (their residual cells hold zeros as well)
            // (kept as channel PAIRS: the arithmetic below is written on two-wide vectors -> v_pk_fma / v_pk_mul / v_pk_add)
            f32x2 sc[2], bi[2], psc[2], psh[2], hw[2];
            {
                const float4 s4 = *reinterpret_cast<const float4*>(wscale + cov0);
                sc[0] = (f32x2){s4.x, s4.y}; sc[1] = (f32x2){s4.z, s4.w};
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (has_bias) b4 = *reinterpret_cast<const float4*>(a.bias + co0);
                bi[0] = (f32x2){b4.x, b4.y}; bi[1] = (f32x2){b4.z, b4.w};
                if constexpr (EPI == EPI_RES_POST) {
                    const float4 p4 = *reinterpret_cast<const float4*>(a.post_scale + co0);
                    const float4 q4 = *reinterpret_cast<const float4*>(a.post_shift + co0);
                    psc[0] = (f32x2){p4.x, p4.y}; psc[1] = (f32x2){p4.z, p4.w};
                    psh[0] = (f32x2){q4.x, q4.y}; psh[1] = (f32x2){q4.z, q4.w};
                }
                if constexpr (EPI == EPI_HEAD) {
                    const float4 h4 = *reinterpret_cast<const float4*>(a.head_w + co0);
                    hw[0] = (f32x2){h4.x, h4.y}; hw[1] = (f32x2){h4.z, h4.w};
                }
            }
            const f32x2 slope2 = {slope, slope};
            // descriptors of the fragment's cells from this tile's plane (+ the sub-pixel parity shift) on: hi plane, lo = + plane_out.
            // 32-bit scalar arithmetic but for the final pointer (the host keeps two cell planes below 4 GiB)
            __amdgpu_buffer_rsrc_t srd_oh, srd_ol;
            if constexpr (EPI != EPI_HEAD && EPI != EPI_PLAIN_F32) {
                const unsigned ex16 = zo16 + (unsigned)(sy * a.Wfull + sx) * 16u;
                const unsigned nrec = ncell > 0 ? (unsigned)ncell * cp16_o - ex16 : 0u;
                const unsigned char* ob8 = out8 + ((size_t)((unsigned)cell0 * (unsigned long long)cp16_o) + ex16);
                srd_oh = make_srd(ob8, nrec);
                srd_ol = make_srd(ob8 + pl16_o, nrec);
            }
            if constexpr (EPI == EPI_POOL) {
                static_assert(EPI != EPI_POOL || (C::RPW % 2 == 0 && C::D == 1), "pooling pairs tile rows inside a wave");
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    if ((n / NFC) % 2 != 0) continue;                 // the even row of each pair handles the pair
                    f32x2 v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 t0 = (f32x2){acc[m][n][2 * h], acc[m][n][2 * h + 1]} * sc[h] + bi[h];
                        f32x2 t1 = (f32x2){acc[m][n + NFC][2 * h], acc[m][n + NFC][2 * h + 1]} * sc[h] + bi[h];
                        t0 = __builtin_elementwise_max(t0, t0 * slope2);
                        t1 = __builtin_elementwise_max(t1, t1 * slope2);
                        const f32x2 t = __builtin_elementwise_max(t0, t1);                       // rows y, y + 1
                        v[h] = __builtin_elementwise_max(t, (f32x2){__shfl_xor(t[0], 1, 64), __shfl_xor(t[1], 1, 64)});   // columns x, x + 1
                    }
                    unsigned h0, l0, h1, l1;
                    split2m(v[0], h0, l0);
                    split2m(v[1], h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    bigacc = __builtin_elementwise_max(bigacc, __builtin_bit_cast(u16x2, h0 & okmask[n]));
                    bigacc = __builtin_elementwise_max(bigacc, __builtin_bit_cast(u16x2, h1 & okmask[n]));
                    __builtin_amdgcn_raw_buffer_store_b64(hi, srd_oh, (int)ovo[n], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(lo, srd_ol, (int)ovo[n], 0, 0);
                }
                continue;
            }
            unsigned keep_h[2] = {0u, 0u}, keep_l[2] = {0u, 0u};     // WIDE: the split halves of the pair's first fragment
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                if constexpr (WIDE && RESID) {
                    if ((n & 1) == 0) { unswap_res(rh, n >> 1); unswap_res(rl, n >> 1); }
                }
                f32x2 v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // acc * scale + (bias + residual): the residual's halves enter through v_fma_mix_f32 (no conversions), the
                    // FMA comes last (an inline-asm result in front of the max would be canonicalised first: v_max x, x)
                    f32x2 addend = bi[h];
                    if constexpr (RESID)
                        addend = add_halves(add_halves(addend, h ? rl[n].y : rl[n].x), h ? rh[n].y : rh[n].x);   // (bias + lo) + hi
                    v[h] = (f32x2){acc[m][n][2 * h], acc[m][n][2 * h + 1]} * sc[h] + addend;
                    if constexpr (EPI == EPI_RES_POST) v[h] = v[h] * psc[h] + psh[h];
                    v[h] = __builtin_elementwise_max(v[h], v[h] * slope2);   // v > 0 ? v : v * slope for every slope <= 1 (host: no others here)
                    if constexpr (EPI == EPI_HEAD) v[h] *= hw[h];
                }
                if constexpr (EPI == EPI_HEAD) {
                    const f32x2 t = v[0] + v[1];
                    hsum[n] += t[0] + t[1];
                } else if constexpr (EPI == EPI_PLAIN_F32) {
                    // fp32 planes [Cout][D][H][W]: four channel planes per lane
                    const size_t cs = cplane_out;
                    float* const fb = a.out_f32 + ((size_t)co0 * cplane_out + zoff_out + (unsigned)(sy * a.Wfull + sx)) + opix[n];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (okv[n] && co0 + r < a.Cout) fb[r * cs] = v[r >> 1][r & 1];
                } else {
                    unsigned h0, l0, h1, l1;
                    split2m(v[0], h0, l0);
                    split2m(v[1], h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    bigacc = __builtin_elementwise_max(bigacc, __builtin_bit_cast(u16x2, h0 & okmask[n]));
                    bigacc = __builtin_elementwise_max(bigacc, __builtin_bit_cast(u16x2, h1 & okmask[n]));
                    if constexpr (WIDE) {
                        if ((n & 1) == 0) {
                            keep_h[0] = h0; keep_h[1] = h1; keep_l[0] = l0; keep_l[1] = l1;
                        } else {
                            const auto sh0 = __builtin_amdgcn_permlane16_swap(keep_h[0], h0, false, false);
                            const auto sh1 = __builtin_amdgcn_permlane16_swap(keep_h[1], h1, false, false);
                            const auto sl0 = __builtin_amdgcn_permlane16_swap(keep_l[0], l0, false, false);
                            const auto sl1 = __builtin_amdgcn_permlane16_swap(keep_l[1], l1, false, false);
                            const u32x4 chi = {sh0[0], sh1[0], sh0[1], sh1[1]}, clo = {sl0[0], sl1[0], sl0[1], sl1[1]};
                            __builtin_amdgcn_raw_buffer_store_b128(chi, srd_oh, (int)ovw[n >> 1], 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(clo, srd_ol, (int)ovw[n >> 1], 0, 0);
                        }
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(hi, srd_oh, (int)ovo[n], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(lo, srd_ol, (int)ovo[n], 0, 0);
                    }
                }
            }
        }
        big = big || bigacc[0] >= 0x7c00 || bigacc[1] >= 0x7c00;
    }  // co-group loop
    if (!(PERSIST && has_next)) break;
    // ---- the workgroup's next tile: its first chunk and weight stage are in the LDS already
    tile_L += tile_stride;
    set_tile_linear(tile_L);
    par_in ^= a.n_chunks & 1;
    par_w ^= n_wstages & 1;
    prefetched = true;
    }  // tile loop

    if constexpr ((ABL & 2048) != 0) {
        if (threadIdx.x == 0) {
            unsigned long long* ctr = reinterpret_cast<unsigned long long*>(a.flag) + 1;
            atomicAdd(ctr, __builtin_readcyclecounter() - probe_c0);
            atomicAdd(ctr + 1, __builtin_amdgcn_s_memrealtime() - probe_r0);
            atomicAdd(ctr + 2, probe_pro);
            atomicAdd(ctr + 3, probe_loop);
        }
        if constexpr ((ABL & 4096) != 0) {
            if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == C::WAVES / 2)) {
                unsigned long long* ctr = reinterpret_cast<unsigned long long*>(a.flag) + 5 + (wave == 0 ? 0 : 4);
                for (int i = 0; i < 4; ++i) atomicAdd(ctr + i, seg[i]);
            }
        }
    }
    if constexpr (EPI == EPI_HEAD) {
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int l15 = tid_e & 15, l4 = (tid_e >> 4) & 3;
        const int wave = __builtin_amdgcn_readfirstlane(tid_e >> 6);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const int trow = wave * C::RPW + n / NFC;
            const int oy = y0 + trow * D;
            const int ox = x0 + (n % NFC) * 16 + l15;
            float h = hsum[n];
            h += __shfl_xor(h, 16, 64);
            h += __shfl_xor(h, 32, 64);
            if (l4 == 0 && oy < a.wy1 && ox < a.wx1) a.head_out[((size_t)oz * a.Hout + oy) * a.Wout + ox] = h + a.head_b;
        }
    } else if constexpr (EPI != EPI_PLAIN_F32) {
        if (__any(big) && lane == 0) atomicOr(a.flag, 1u);
    }
}

template <class C, int EPI, int ABL = 0, int MODE = 3>
__global__ __launch_bounds__(C::THREADS, C::WGS_PER_CU) void conv_split_kernel(const SplitArgs a) {
    conv_split_body<C, EPI, ABL, MODE>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// BATCHED launch: the same layer of up to SPLIT_MULTI_MAX independent images (the patches of a patched denoise, the tiles of a
// tomogram: denoise.py:299-323, 340-377) in ONE grid.  The deep levels of a U-Net are 16-tile launches on a 256-CU chip (20 - 60
// us each whatever they compute, ~400 of them per micrograph); entry j's workgroups are the linear ids [first[j], first[j+1])
// (first[] rounded to multiples of 8: the XCD of a workgroup stays its tile index mod 8; the gap workgroups exit), laid out
// x-fastest over its own (tiles_x, tiles_y, z) grid.  The whole table travels in the kernel-argument segment and is read
// through the scalar cache like the arguments of a single launch: the entry index is wave-uniform.
enum { SPLIT_MULTI_MAX = 8 };
struct SplitMulti {
    unsigned n, first[SPLIT_MULTI_MAX + 1];
    unsigned pad_[6];                            // (the entries start 64-byte aligned)
    SplitArgs a[SPLIT_MULTI_MAX];
};
static_assert(sizeof(SplitMulti) <= 4096, "kernel-argument segment");

template <class C, int EPI, int MODE>
__global__ __launch_bounds__(C::THREADS, C::WGS_PER_CU) void conv_split_multi_kernel(const SplitMulti) {
    // (addressed through the segment pointer, not through the parameter: indexing a by-value aggregate with a run-time index
    // makes the compiler copy it to scratch)
    typedef const __attribute__((address_space(4))) SplitMulti* multi_ptr_t;
    const multi_ptr_t m = (multi_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned b = blockIdx.x;
    unsigned j = 0;                              // (first[] is monotone and first[i >= n] = the grid size: no bound check)
#pragma unroll
    for (unsigned i = 1; i < SPLIT_MULTI_MAX; ++i) j += b >= m->first[i] ? 1u : 0u;
    const unsigned l = b - m->first[j];
    const SplitArgs& a = *(const SplitArgs*)&m->a[j];
    const unsigned gx = (unsigned)a.tiles_x, gy = (unsigned)a.tiles_y, gxy = gx * gy;
    const unsigned bz = l / gxy, rem = l - bz * gxy, by = rem / gx;
    if (l >= (unsigned)a.n_tiles) return;        // (padding workgroups between two entries)
    conv_split_body<C, EPI, 0, MODE>(a, rem - by * gx, by, bz, gx, gy);
}

}  // namespace tpz
