// Internal declarations of the libtopaz_hip.so runtime (NOT part of the C-ABI: include/topaz_hip.h is).
// The runtime is split into translation units by concern:
//   rt_core.hip     kernel registries, the context (streams, workspace pools, patch lanes, the batched-launch recorder, the
//                   HIP-event profiler), its setters, the debug switches
//   rt_load.hip     model loading: kernel choice per layer, weight packing (fp32 and 2xf16 forms), per-parity decoder forms,
//                   folded projections, zero-padded widths, the bias arena
//   rt_exec.hip     the layer-program executor: launchers, per-layer drivers, the backward walk of the windows (need_regions)
//   rt_forward.hip  scoring drivers (range-scaled pass, internal tiling) and the single-op entry points
//   rt_denoise.hip  the patched 2-D and tiled 3-D denoising drivers (batched passes over the lanes)
//   rt_stats.hip    mean / std, GMM fit, affine, normalise
//   rt_stage.hip    the staging ring and the host-pointer entry points
//   rt_nms.hip      the NMS driver
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <string>
#include <vector>

#include "../../include/topaz_hip.h"
#include "conv_registry.h"
#include "conv_split_registry.h"
#include "conv_rw.h"
#include "kernels_misc.h"

using namespace tpz;

// ---- debug / A-B switches ----------------------------------------------------------------------------------------------
// Every switch below changes how a job computes or is scheduled.  They exist for tools/ and tests/ (same-process A/B legs,
// bit-identity checks); a stray variable in a user's environment must not change a job, so the environment is consulted by ONE
// function, debug_env(), and only when TPZ_DEBUG=1 is set.  It is read when a context is created (the ctx keeps that copy:
// tpz_ctx::dbg) and again at every model load.  The settings that matter outside tests have a tpz_ctx_set_* entry point.
struct DebugEnv {
    bool on = false;             // TPZ_DEBUG=1
    bool no_phase = false;       // TPZ_NO_PHASE     keep the fused upsample+concat loader for every decoder layer
    bool exact_fp32 = false;     // TPZ_EXACT_FP32   every network on the fp32 MFMA kernels (tpz_ctx_set_exact)
    bool no_issuer = false;      // TPZ_NO_ISSUER    every wave issues its own share of the per-step LDS-DMA
    bool no_lanes = false;       // TPZ_NO_LANES     patches / tiles of an image on one stream only (tpz_ctx_set_lanes)
    bool no_roi = false;         // TPZ_NO_ROI       every layer of a patch computes its whole tensor (tpz_ctx_set_roi)
    bool no_persist = false;     // TPZ_NO_PERSIST   no persistent workgroups (tpz_ctx_set_persist)
    bool trace_host = false;     // TPZ_TRACE_HOST   host time of the recording / issuing phases of a batched pass, on stderr
    bool no_range = false;       // TPZ_NO_RANGE     no range-scaled scoring pass (tpz_ctx_set_range)
    bool no_raster = false;      // TPZ_NO_RASTER    row-major tile raster (tpz_ctx_set_raster)
    bool no_srcmajor = false;    // TPZ_NO_SRCMAJOR  plane-major virtual cells for two-source 3-D layers (MODE 3)
    bool no_valu_last = false;   // TPZ_NO_VALU_LAST the 1-output-channel last conv as a column kernel + shift-sum (round 4)
    bool no_rw = false;          // TPZ_NO_RW        no weights-resident kernel for the 3x3 32 -> 32 layers (tpz_ctx_set_rw)
    bool no_pool3d = false;      // TPZ_NO_POOL3D    no fused max-pool epilogue on 3-D encoder convs
    bool no_fold = false;        // TPZ_NO_FOLD      ResidA 1x1 projections as layers of their own
    bool no_widen = false;       // TPZ_NO_WIDEN     never load a program zero-padded to multiples of 16 channels
    int batch = -1;              // TPZ_BATCH=n / TPZ_NO_BATCH=1 (0)   images per batched pass (tpz_ctx_set_batch); -1: default
    int lanes = 0;               // TPZ_LANES=n      patch lanes in use (tpz_ctx_set_lanes); 0: default
};
DebugEnv debug_env();

static inline double host_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#ifndef TPZ_N_LANES
#define TPZ_N_LANES 4
#endif
enum { N_LANES = TPZ_N_LANES };      // patch lanes: the most auxiliary streams the patches / tiles of an image alternate on
enum { NMS_BATCH = 4, NMS_VER = 5, NMS_SNAP = 9, NMS_PICKS = 15, NMS_COUNTERS = 16 };

struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops, bytes;  // algorithmic FLOP and HBM bytes of the launch (bytes: read inputs + weights once, write outputs once)
    const void* key;      // identity of the kernel instantiation (its registry name / a static label), nullptr for the rest
};
struct ProfAcc {
    double ms = 0, flops = 0, bytes = 0;
    long long n = 0;
};

// One deferred launch of a batched pass (tpz_ctx::rec): a conv_split launch (ks != nullptr: `a` complete but for the stream,
// a.n_tiles = the workgroups of `grid`) that rec_flush may merge with the same layer's launch of other images, or any other
// launch as a closure over its arguments.
struct RecOp {
    const SplitKernelInfo* ks = nullptr;
    SplitArgs a;
    dim3 grid;
    std::function<hipError_t(hipStream_t)> fn;
    int cls = 2;
    double flops = 0, bytes = 0;
    const void* key = nullptr;
};

struct tpz_ctx {
    int device = 0;
    DebugEnv dbg;                 // the debug switches as they stood when the ctx was created (all off without TPZ_DEBUG=1)
    int n_cus = 256;              // compute units (persistent grids are sized from it)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    struct Buf {
        void* p;
        size_t bytes;
        bool used;
    };
    std::vector<Buf> pool;        // workspace of the ctx stream
    std::vector<Buf>* pool_cur = &pool;
    // Patch lanes: independent patches / tiles of one image are enqueued round-robin on N_LANES auxiliary streams, each
    // with its own workspace pool and reduction scratch, so that the small, latency-bound launches of one patch (the
    // deep U-Net levels: 16-tile grids on 256 CUs) run under the large ones of its neighbour (lanes_begin / lane_enter /
    // lanes_end).  One host thread enqueues everything; nothing synchronises with the host.
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        std::vector<Buf> pool;
        double* d_part = nullptr;
    };
    Lane lanes[N_LANES];
    struct tpz_stage* io_stage = nullptr;     // ring behind the host-pointer entry points (created on first use)
    hipEvent_t lanes_fork = nullptr;
    hipStream_t lanes_saved_stream = nullptr;
    double* lanes_saved_part = nullptr;
    bool lanes_on = false;
    int lanes_live = 2;                       // ... of the lanes_begin in progress
    int n_lanes = 2;                          // lanes in use (<= N_LANES): tpz_ctx_set_lanes(ctx, n), TPZ_LANES
    bool lanes_enabled = true;                // tpz_ctx_set_lanes
    bool roi_enabled = true;                  // tpz_ctx_set_roi: patches compute only what their kept centre depends on
    int persist_mode = 1;                     // tpz_ctx_set_persist: 0 never, 1 large launches (default), 2 every eligible launch
    int persist_wgs = 0;                      // ... workgroups of a persistent grid (0: CUs x workgroups per CU)
    // Batched passes (rec_begin / rec_select / rec_flush): the launches of up to SPLIT_MULTI_MAX independent images (patches of
    // a micrograph, tiles of a tomogram) are RECORDED, image by image, each image on a workspace pool of its own, and then
    // issued layer by layer -- the conv_split launches of the same layer as ONE grid (conv_split_multi_kernel).  The deep
    // levels of a U-Net are 16-tile launches on a 256-CU chip; batched they are 8 x as large and 8 x fewer.
    int batch = (int)tpz::SPLIT_MULTI_MAX;    // images per batch (tpz_ctx_set_batch); 0: off (patch lanes)
    bool rec_on = false;
    double rec_t0 = 0;                        // (TPZ_TRACE_HOST)
    int rec_cur = 0;
    std::vector<RecOp> rec[tpz::SPLIT_MULTI_MAX];
    std::vector<Buf> rec_pools[TPZ_N_LANES][tpz::SPLIT_MULTI_MAX];   // per lane (two batches are in flight at a time) and image
    int rec_lane = 0;
    long long batch_mem = 0;                  // tpz_ctx_set_batch_memory: device bytes a batched pass may take (0: what is free)
    long long n_launches = 0;                 // kernel launches issued (tpz_prof_launches)
    double* d_part = nullptr;     // reduction partials
    float* d_nrm = nullptr;       // ring of float[4] normalisation parameter blocks
    int nrm_next = 0;
    unsigned int* d_counters = nullptr;     // NMS_COUNTERS entries (nms_common)
    float* d_zeros = nullptr;     // 256 B of zeros: DMA source of padded / out-of-image elements
    // range-scaled scoring pass (tpz_model_forward): every bias-like vector is read `bias_shift` floats further on (the model's
    // scaled copy of its bias arena), the fused head adds no bias (the un-scaling pass does)
    ptrdiff_t bias_shift = 0;
    bool scaled_pass = false;
    unsigned* d_absmax = nullptr; // exponent histogram of launch_range_fit (256 words, kept zeroed)
    bool range_scaling = true;                // tpz_ctx_set_range
    bool raster = true;                       // tpz_ctx_set_raster: patch raster of the 8-wave tiles' grids
    bool rw_enabled = true;                   // tpz_ctx_set_rw: the weights-resident kernel for 3x3 32 -> 32 layers (conv_rw.h)
    // internal tiling of tpz_model_forward (run_image): 2-D images above tile_limit_px pixels are scored in tile_size^2 tiles
    long long tile_limit_px = 40LL << 20;
    int tile_size = 4096;
    unsigned* d_flag = nullptr;   // f16-range overflow flag of the 2xf16 path
    unsigned* h_flag = nullptr;   // pinned copy
    bool exact = false;           // fp32 kernels only (tpz_ctx_set_exact)
    // K-loop schedules of the 2xf16 kernels (conv_split.h SplitStep), built on first use per (kernel, layer shape) and kept
    // on the device for the life of the ctx: (kernel, key) -> device table
    struct SplitPlan {
        const SplitKernelInfo* ks;
        SplitPlanKey key;
        SplitStep* d;
        bool next_ok;             // holds a complete next-tile fetch: the persistent kernel may run this layer
    };
    std::vector<SplitPlan> split_plans;
    // profiling
    int prof = 0;                 // 0 off, 1 every launch, 2 conv launches of >= 20 GFLOP only (cheap enough for timed runs)
    bool prof_open = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> free_events;
    std::vector<std::pair<const void*, ProfAcc>> per_kernel;   // conv_mfma instantiations
    double acc_ms[4] = {0, 0, 0, 0};
    long long acc_n[4] = {0, 0, 0, 0};
    double acc_flops[4] = {0, 0, 0, 0};
};

static const int PART_BLOCKS = 1024;
static const int NRM_RING = 4096;

int fail(tpz_ctx* ctx, const char* fmt, ...);

#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// ---- rt_core.hip
void* pool_alloc(tpz_ctx* ctx, size_t bytes);
void pool_release(tpz_ctx* ctx, void* p);
float* next_nrm(tpz_ctx* ctx);
int lanes_begin(tpz_ctx* ctx);
void lane_enter(tpz_ctx* ctx, int k);
int lanes_end(tpz_ctx* ctx);
void prof_begin(tpz_ctx* ctx, int cls, double flops, const void* key = nullptr, double bytes = 0);
void prof_end(tpz_ctx* ctx);
void prof_flush(tpz_ctx* ctx);
int rec_begin(tpz_ctx* ctx, int batch_no = 0);
void rec_select(tpz_ctx* ctx, int i);
void rec_abort(tpz_ctx* ctx);
int rec_flush(tpz_ctx* ctx);
const std::string& last_global_error();
void restore_errors(tpz_ctx* ctx, const std::string& ctx_err, const std::string& global_err);

// ---- launches: issued at once on the ctx stream or, in a batched pass, recorded for rec_flush
// fn(stream) launches the kernel(s); cls / flops / key / bytes label it for the profiler
template <class F>
static hipError_t enqueue(tpz_ctx* ctx, int cls, double flops, const void* key, double bytes, F&& fn) {
    if (ctx->rec_on) {
        RecOp op;
        op.fn = std::forward<F>(fn);
        op.cls = cls; op.flops = flops; op.key = key; op.bytes = bytes;
        ctx->rec[ctx->rec_cur].push_back(std::move(op));
        return hipSuccess;
    }
    prof_begin(ctx, cls, flops, key, bytes);
    const hipError_t e = fn(ctx->stream);
    prof_end(ctx);
    ++ctx->n_launches;
    return e;
}
template <class F>
static hipError_t enqueue(tpz_ctx* ctx, F&& fn) { return enqueue(ctx, 2, 0.0, nullptr, 0.0, std::forward<F>(fn)); }

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct LayerRT {
    tpz_layer L;
    const ConvKernelInfo* ki = nullptr;   // nullptr -> direct kernel
    int n_cog = 1, n_chunks = 1, cog_inner = 1;
    int c1 = 0, c2 = 0;                   // channels of the first / second source
    float* d_wpk = nullptr;               // packed (MFMA) or raw (direct) weights
    float* d_bias = nullptr;
    float* d_post_scale = nullptr;
    float* d_post_shift = nullptr;
    float* d_head_w = nullptr;
    float head_b = 0.f;
    float bias0 = 0.f;                    // bias of output channel 0 (host copy, for the 1-output-channel convs)
    // 2xf16 path (prepare_split): kernel, packed hi/lo weights, per-channel 2^-s; or the stem that feeds it
    const SplitKernelInfo* ks = nullptr;
    const ConvKernelInfo* ki_stem_split = nullptr;
    void* d_wsplit = nullptr;
    float* d_wscale = nullptr;
    int s_n_cog = 1, s_n_chunks = 1;
    // column-kernel forms (prepare_split): a 1-channel stem as an 8*ncell-channel conv over an x-shifted copy of the
    // image (kx taps as input channels), a 1-output-channel conv as k virtual output channels + a shifted sum
    const SplitKernelInfo* ks_stem = nullptr;
    const SplitKernelInfo* ks_last = nullptr;
    float* d_wlast = nullptr;                  // ... or (k = 3 / 5, few input channels) the vector-ALU stencil conv_cout1_split_kernel:
                                               // its weights [kz][cell][kx][ky][8] fp32
    // 3x3 32 -> 32 layers (dilation 1 / 2 / 4) of the 32-unit detectors: the weights-resident persistent kernel (conv_rw.h),
    // its weights packed with all 4 cells of a tap per step
    void* d_w_rw = nullptr;
    float* d_ws_rw = nullptr;
    const SplitKernelInfo* ks_pool = nullptr;  // twin of ks / ks_stem with the following 2x2 max-pool fused (EPI_POOL)
    // ResidA blocks that change width, y = [bn1](conv1(t) + proj(h)) (resnet.py:185-202): on the 2xf16 path the 1x1 projection is
    // FOLDED into conv1's K loop (SplitArgs::fold_cells) -- the projection layer is then skipped (folded_into = index of conv1)
    // and conv1 runs ks_fold (one-step stages, plain epilogue) over its own source + slot fold_src, eval-BN folded into weights
    int folded_into = -1;
    int fold_src = -1, fold_cells = 0, f_n_cog = 1, f_n_chunks = 1;
    const SplitKernelInfo* ks_fold = nullptr;
    void* d_wfold = nullptr;
    float* d_wscale_fold = nullptr;
    float* d_bias_fold = nullptr;
    // 2xf16 twin of the phase decomposition: the skip-source part runs first (stem kernel storing split cells
    // when the skip is the 1-channel image, else a plain split kernel), then one split kernel per output parity
    // adds itself in place through the residual epilogue and applies the activation
    struct SplitPhase {
        bool valid = false;
        const SplitKernelInfo* ks_low = nullptr;       // k1-tap kernel, EPI_RES, lattice output
        const SplitKernelInfo* ks_skip = nullptr;      // k-tap kernel over a multi-channel skip source, EPI_PLAIN
        const ConvKernelInfo* ki_skip_stem = nullptr;  // 1-channel skip source: fp32 CIN1 kernel, EPI_SPLIT
        int n_cog_low = 1, n_chunks_low = 1, n_cog_skip = 1, n_chunks_skip = 1;
        const SplitKernelInfo* ks_sub = nullptr;       // 5x5: all parities as 4*cout virtual channels of one 3x3 conv
        int n_cog_sub = 1;
        bool sub_with_skip = false;            // ... the 1-channel skip source folded in as 4 space-to-depth channels
        bool low_with_skip = false;            // 3x3(x3): the same fold into the per-parity kernels (one more cell)
        bool srcmajor = false;                 // ... in 3-D with the virtual cells ordered source-major (conv_split.h MODE 11)
        const SplitKernelInfo* ks_low_plain = nullptr;
        void* d_w_low = nullptr;               // the packs of all parities, w_phase_bytes apart
        float* d_ws_low = nullptr;             // [parity][cout]
        size_t w_phase_bytes = 0;
        void* d_w_skip = nullptr;
        float* d_ws_skip = nullptr;
    } sphase;
    // phase decomposition (prepare_phases): the first source arrives 2x nearest-upsampled
    struct Phase {
        bool valid = false;
        int c1 = 0, c2 = 0, k1 = 0;
        const ConvKernelInfo* ki_low = nullptr;    // k1-tap kernel over the low-resolution source, EPI_PLAIN
        const ConvKernelInfo* ki_skip = nullptr;   // k-tap kernel over the skip source, EPI_RES (in place)
        int n_cog_low = 1, n_chunks_low = 1, n_cog_skip = 1, n_chunks_skip = 1;
        float* d_w_low[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        float* d_w_skip = nullptr;
    } phase;
};

struct tpz_model {
    tpz_ctx* ctx = nullptr;
    DebugEnv dbg;                         // the debug switches read when this model was loaded
    std::vector<LayerRT> layers;
    int n_slots = 0;
    std::vector<int> last_use;
    std::vector<void*> dev_allocs;
    // bias-like vectors (conv biases, folded biases, eval-BN shifts) of all layers in one arena + a scratch copy of the same
    // size that a range-scaled pass fills with 2^-s times the originals (tpz_model_forward)
    float* d_bias_arena = nullptr;
    float* d_bias_scaled = nullptr;
    size_t n_bias_arena = 0;
    int n_conv = 0, n_conv_split = 0;     // convolution layers; those with a 2xf16 kernel (prepare_split)
    std::string off_path;                 // ... the others, "#layer KxK dD cin->cout, ..."
    bool split_ok = false;                // at least one layer has a 2xf16 kernel: the program runs in split mode
    bool widened = false;                 // the program was loaded with its widths zero-padded to multiples of 16 (widen_program)
    long long n_split = 0, n_fallback = 0;
};

// a rectangle of a 2-D tensor (planes [z0, z1) of a 3-D one: a box); on = false: the whole tensor
struct Rect {
    int y0 = 0, x0 = 0, y1 = 0, x1 = 0;
    bool on = false;
    int z0 = 0, z1 = 1;
    void unite(const Rect& r) {
        if (!r.on) return;
        if (!on) { *this = r; return; }
        y0 = std::min(y0, r.y0); x0 = std::min(x0, r.x0); y1 = std::max(y1, r.y1); x1 = std::max(x1, r.x1);
        z0 = std::min(z0, r.z0); z1 = std::max(z1, r.z1);
    }
    long long area() const { return (long long)(y1 - y0) * (x1 - x0); }
};

struct Slot {
    Rect need;                // the part of the tensor that anything reads (need_regions); the producer computes just that
    float* p = nullptr;
    int C = 0, D = 1, H = 0, W = 0;
    long long cs = 0, ps = 0;
    int pitch = 0;
    bool owned = false;
    bool set = false;
    bool split = false;       // p holds split f16 cells (split_fmt.h) instead of fp32 planes
    bool pooled = false;      // the producing conv already applied the max-pool that follows it (EPI_POOL)
    float* alt = nullptr;     // the same tensor converted to the other format for a consumer that needs it
};

static void set_dense(Slot& s, float* p, int C, int D, int H, int W) {
    s.p = p; s.C = C; s.D = D; s.H = H; s.W = W;
    s.pitch = W; s.ps = (long long)H * W; s.cs = s.ps * D;
    s.set = true;
}

// a bias-like vector as the current pass reads it (the scaled copy in a range-scaled pass)
static inline const float* bias_view(const tpz_ctx* ctx, const float* p) { return p ? p + ctx->bias_shift : nullptr; }

// ---- cross-TU functions (defined in the file named)
// rt_load.hip
static inline size_t chan_pad(size_t n) { return (n + 127) / 128 * 128 + 128; }     // per-channel vectors: whole 128-channel tiles + one
static inline int phase_pad(int k, int p) { return (k / 2 - p + 1) / 2; }
int upload(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out);
int upload_chan(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out);
void pack_weights_split(const SplitKernelInfo& ki, const float* w, int cout, int cin, int n_cog, int n_chunks,
                        std::vector<uint16_t>& out, std::vector<float>& wscale_inv, const float* wp = nullptr, int cin_b = 0,
                        const float* mul = nullptr);
const SplitKernelInfo* pick_split(int k, int dil, int cout, int epi, int kx = 0);
int model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
               const std::vector<int>& preset_chan, tpz_model** out);
// rt_exec.hip
int run_conv_split(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, const Slot* sres, Slot& dst, const Slot* s2 = nullptr,
                   bool pooled = false, const Slot* fold = nullptr);
int run_program(tpz_model* m, std::vector<Slot>& slots, float* d_out, const float* d_nrm, bool split = false,
                const Rect* keep = nullptr);
