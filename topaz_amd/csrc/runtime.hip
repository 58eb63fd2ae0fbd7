// libtopaz_hip.so runtime: context, weight packing, layer-program executor, denoise / NMS drivers
// and the C-ABI declared in include/topaz_hip.h.
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

// ------------------------------------------------------------------------------------------------
// kernel registry
// ------------------------------------------------------------------------------------------------
namespace tpz {
static std::vector<ConvKernelInfo>& registry() {
    static std::vector<ConvKernelInfo> r;
    return r;
}
void register_conv(const ConvKernelInfo& info) { registry().push_back(info); }
const ConvKernelInfo* find_conv(int dims, int K, int D, int MT, bool cin1, int epi) {
    for (const auto& k : registry())
        if (k.dims == dims && k.K == K && k.D == D && k.MT == MT && k.cin1 == (cin1 ? 1 : 0) && k.epi == epi) return &k;
    return nullptr;
}
static std::vector<SplitKernelInfo>& split_registry() {
    static std::vector<SplitKernelInfo> r;
    return r;
}
void register_split(const SplitKernelInfo& info) { split_registry().push_back(info); }
const SplitKernelInfo* find_split(int K, int D, int MT, int epi, int KX, int sps) {
    if (KX <= 0) KX = K;
    // several instantiations of one shape may differ in the steps per stage: the most steps (fewest barriers) unless the
    // caller needs a particular form (sps > 0: the folded 1x1 projection exists for one-step stages only)
    const SplitKernelInfo* best = nullptr;
    for (const auto& k : split_registry())
        if (k.K == K && k.KX == KX && k.D == D && k.MT == MT && k.epi == epi && (sps <= 0 || k.SPS == sps) &&
            (!best || k.SPS > best->SPS)) best = &k;
    return best;
}
}  // namespace tpz

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static std::string g_last_error;
// TPZ_NO_PHASE=1 keeps the fused upsample+concat loader for every decoder layer (A/B switch for tests and tuning)
static const bool g_no_phase = getenv("TPZ_NO_PHASE") != nullptr;
// TPZ_EXACT_FP32=1 keeps every network on the fp32 MFMA kernels (the 2xf16 path of conv_split.h is never taken)
static const bool g_exact_fp32 = getenv("TPZ_EXACT_FP32") != nullptr;
// TPZ_NO_ISSUER=1: every wave issues its own share of the per-step LDS-DMA (A/B switch for tuning, see launch_split)
static const bool g_no_issuer = getenv("TPZ_NO_ISSUER") != nullptr;
static const bool g_no_lanes = getenv("TPZ_NO_LANES") != nullptr;      // patches / tiles of an image on one stream only
// TPZ_NO_ROI=1: every layer of a patch computes its whole tensor (A/B switch; see need_regions)
static const bool g_no_roi = getenv("TPZ_NO_ROI") != nullptr;
static const bool g_persist = getenv("TPZ_NO_PERSIST") == nullptr;      // persistent workgroups for the large plain conv_split launches
// TPZ_BATCH=n: the same layer of n patches / tiles of an image in one launch (0 / TPZ_NO_BATCH=1: patch lanes instead)
// TPZ_TRACE_HOST=1: host time of the recording and issuing phases of a batched pass, on stderr
static const bool g_trace_host = getenv("TPZ_TRACE_HOST") != nullptr;
static double host_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int env_batch() {
    if (getenv("TPZ_NO_BATCH")) return 0;
    const char* e = getenv("TPZ_BATCH");
    const int n = e ? atoi(e) : (int)tpz::SPLIT_MULTI_MAX;
    return n < 0 ? 0 : n > (int)tpz::SPLIT_MULTI_MAX ? (int)tpz::SPLIT_MULTI_MAX : n;
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
    bool lanes_enabled = !g_no_lanes;         // tpz_ctx_set_lanes
    bool roi_enabled = !g_no_roi;             // tpz_ctx_set_roi: patches compute only what their kept centre depends on
    int persist_mode = g_persist ? 1 : 0;     // tpz_ctx_set_persist: 0 never, 1 large launches (default), 2 every eligible launch
    int persist_wgs = 0;                      // ... workgroups of a persistent grid (0: CUs x workgroups per CU)
    // Batched passes (rec_begin / rec_select / rec_flush): the launches of up to SPLIT_MULTI_MAX independent images (patches of
    // a micrograph, tiles of a tomogram) are RECORDED, image by image, each image on a workspace pool of its own, and then
    // issued layer by layer -- the conv_split launches of the same layer as ONE grid (conv_split_multi_kernel).  The deep
    // levels of a U-Net are 16-tile launches on a 256-CU chip; batched they are 8 x as large and 8 x fewer.
    int batch = env_batch();                  // images per batch (tpz_ctx_set_batch); 0: off (patch lanes)
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
    bool range_scaling = getenv("TPZ_NO_RANGE") == nullptr;       // tpz_ctx_set_range
    bool raster = getenv("TPZ_NO_RASTER") == nullptr;             // tpz_ctx_set_raster: patch raster of the 8-wave tiles' grids
    bool rw_enabled = true;                   // tpz_ctx_set_rw: the weights-resident kernel for 3x3 32 -> 32 layers (conv_rw.h)
    // internal tiling of tpz_model_forward (run_image): 2-D images above tile_limit_px pixels are scored in tile_size^2 tiles
    long long tile_limit_px = 40LL << 20;
    int tile_size = 4096;
    unsigned* d_flag = nullptr;   // f16-range overflow flag of the 2xf16 path
    unsigned* h_flag = nullptr;   // pinned copy
    bool exact = g_exact_fp32;    // fp32 kernels only
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

static int fail(tpz_ctx* ctx, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return 1;
}

#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

static void* pool_alloc(tpz_ctx* ctx, size_t bytes) {
    if (bytes == 0) bytes = 16;
    std::vector<tpz_ctx::Buf>& pool = *ctx->pool_cur;
    int best = -1;
    for (int i = 0; i < (int)pool.size(); ++i) {
        auto& b = pool[i];
        if (!b.used && b.bytes >= bytes && (best < 0 || b.bytes < pool[best].bytes)) best = i;
    }
    if (best >= 0) {
        pool[best].used = true;
        return pool[best].p;
    }
    void* p = nullptr;
    // round up so slightly larger requests can reuse the buffer
    size_t rounded = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    if (hipMalloc(&p, rounded) != hipSuccess) {
        // drop unused cached buffers and retry: first this pool's, then those of EVERY pool of the ctx (the pools of a batched
        // pass -- up to lanes x 8 of them --, the lane pools, the ctx pool: a pass with other tile shapes, or a large frame on
        // the ctx stream after a batched pass filled the device, finds its memory cached elsewhere)
        auto trim = [](std::vector<tpz_ctx::Buf>& pl) {
            for (auto it = pl.begin(); it != pl.end();) {
                if (!it->used) { (void)hipFree(it->p); it = pl.erase(it); }
                else ++it;
            }
        };
        (void)hipGetLastError();
        trim(pool);
        if (hipMalloc(&p, rounded) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();          // (buffers released by passes still in flight on the lanes)
            trim(ctx->pool);
            for (auto& ln : ctx->lanes) trim(ln.pool);
            for (auto& lane_pools : ctx->rec_pools)
                for (auto& pl : lane_pools) trim(pl);
            if (hipMalloc(&p, rounded) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
    }
    pool.push_back({p, rounded, true});
    return p;
}
static void pool_release(tpz_ctx* ctx, void* p) {
    for (auto& b : *ctx->pool_cur)
        if (b.p == p) { b.used = false; return; }
}

static float* next_nrm(tpz_ctx* ctx) {
    if (ctx->nrm_next >= NRM_RING) {
        if (ctx->lanes_on) (void)hipDeviceSynchronize();     // the other lane may still read blocks of this ring
        else (void)hipStreamSynchronize(ctx->stream);
        ctx->nrm_next = 0;
    }
    return ctx->d_nrm + 4 * (ctx->nrm_next++);
}

// ---- patch lanes (tpz_ctx::Lane)
static int lanes_begin(tpz_ctx* ctx) {
    if (!ctx->lanes_enabled || ctx->lanes_on) return 0;
    if (!ctx->lanes_fork) {
        if (hipEventCreateWithFlags(&ctx->lanes_fork, hipEventDisableTiming) != hipSuccess) return fail(ctx, "hipEventCreate failed");
        for (auto& ln : ctx->lanes) {
            if (hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ln.done, hipEventDisableTiming) != hipSuccess ||
                hipMalloc((void**)&ln.d_part, 2 * PART_BLOCKS * sizeof(double)) != hipSuccess)
                return fail(ctx, "patch lanes: stream / event / scratch creation failed");
        }
    }
    // the lanes start after everything already queued on the ctx stream (the input image, the cleared overflow flag)
    HIPCHK(ctx, hipEventRecord(ctx->lanes_fork, ctx->stream));
    for (auto& ln : ctx->lanes) HIPCHK(ctx, hipStreamWaitEvent(ln.stream, ctx->lanes_fork, 0));
    ctx->lanes_saved_stream = ctx->stream;
    ctx->lanes_saved_part = ctx->d_part;
    ctx->lanes_on = true;
    ctx->lanes_live = ctx->n_lanes;
    return 0;
}
static void lane_enter(tpz_ctx* ctx, int k) {
    if (!ctx->lanes_on) return;
    tpz_ctx::Lane& ln = ctx->lanes[k % ctx->lanes_live];
    ctx->stream = ln.stream;
    ctx->pool_cur = &ln.pool;
    ctx->d_part = ln.d_part;
}
static int lanes_end(tpz_ctx* ctx) {
    if (!ctx->lanes_on) return 0;
    ctx->stream = ctx->lanes_saved_stream;
    ctx->pool_cur = &ctx->pool;
    ctx->d_part = ctx->lanes_saved_part;
    ctx->lanes_on = false;
    for (auto& ln : ctx->lanes) {
        HIPCHK(ctx, hipEventRecord(ln.done, ln.stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ln.done, 0));
    }
    return 0;
}

// ---- profiling helpers
static void prof_begin(tpz_ctx* ctx, int cls, double flops, const void* key = nullptr, double bytes = 0) {
    ctx->prof_open = false;
    if (!ctx->prof) return;
    if (ctx->prof == 2 && (cls != 0 || flops < 2e10)) return;
    ctx->prof_open = true;
    ProfRec r;
    r.cls = cls;
    r.flops = flops;
    r.bytes = bytes;
    r.key = key;
    auto get = [&]() {
        hipEvent_t e;
        if (!ctx->free_events.empty()) { e = ctx->free_events.back(); ctx->free_events.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    r.e0 = get();
    r.e1 = get();
    (void)hipEventRecord(r.e0, ctx->stream);
    ctx->recs.push_back(r);
}
static void prof_end(tpz_ctx* ctx) {
    if (!ctx->prof_open || ctx->recs.empty()) return;
    ctx->prof_open = false;
    (void)hipEventRecord(ctx->recs.back().e1, ctx->stream);
}
static void prof_flush(tpz_ctx* ctx) {
    if (ctx->recs.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->recs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        ctx->acc_ms[r.cls] += ms;
        ctx->acc_n[r.cls] += 1;
        ctx->acc_flops[r.cls] += r.flops;
        if (r.key) {
            ProfAcc* a = nullptr;
            for (auto& kv : ctx->per_kernel)
                if (kv.first == r.key) a = &kv.second;
            if (!a) { ctx->per_kernel.push_back({r.key, ProfAcc()}); a = &ctx->per_kernel.back().second; }
            a->ms += ms; a->flops += r.flops; a->bytes += r.bytes; a->n += 1;
        }
        ctx->free_events.push_back(r.e0);
        ctx->free_events.push_back(r.e1);
    }
    ctx->recs.clear();
}

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

// `batch_no`: consecutive batches alternate on the patch lanes (lane_enter: the lane's stream and reduction scratch), so that
// one batch's elementwise launches and small grids run under the other's large ones -- batches of 8 images, two in flight
static int rec_begin(tpz_ctx* ctx, int batch_no = 0) {
    if (g_trace_host) ctx->rec_t0 = host_now_ms();
    for (auto& r : ctx->rec) r.clear();
    lane_enter(ctx, batch_no);                  // (no-op without lanes: everything on the ctx stream)
    ctx->rec_lane = ctx->lanes_on ? batch_no % ctx->lanes_live : 0;
    ctx->rec_on = true;
    ctx->rec_cur = 0;
    ctx->pool_cur = &ctx->rec_pools[ctx->rec_lane][0];
    return 0;
}
// image i of the batch: its launches are recorded in its own list, its tensors come from its own pool (the images of a batch
// run interleaved: nothing of one may alias anything of another)
static void rec_select(tpz_ctx* ctx, int i) {
    ctx->rec_cur = i;
    ctx->pool_cur = &ctx->rec_pools[ctx->rec_lane][i];
}
// leaves a batched pass: whatever is still recorded (an error on the way) is dropped
static void rec_abort(tpz_ctx* ctx) {
    ctx->rec_on = false;
    ctx->pool_cur = &ctx->pool;
    for (auto& r : ctx->rec) r.clear();
}
// Issue everything recorded.  Every list keeps its own order (the dependencies inside an image); across the lists the
// launches are independent, so each round first replays the non-convolution launches at the head of every list and then takes
// the conv_split launch at the head of the first unfinished list together with every other list's head that is the same
// kernel in the same mode with the same K-loop plan: one grid.
static int rec_flush(tpz_ctx* ctx) {
    ctx->rec_on = false;
    ctx->pool_cur = &ctx->pool;
    const double t_flush0 = g_trace_host ? host_now_ms() : 0.0;
    long long n_issued = 0;
    const int n = (int)tpz::SPLIT_MULTI_MAX;
    size_t cur[tpz::SPLIT_MULTI_MAX] = {};
    int rc = 0;
    for (;;) {
        bool any = false;
        for (int i = 0; i < n && !rc; ++i) {
            auto& L = ctx->rec[i];
            while (cur[i] < L.size() && !L[cur[i]].ks && !rc) {
                RecOp& op = L[cur[i]++];
                prof_begin(ctx, op.cls, op.flops, op.key, op.bytes);
                const hipError_t e = op.fn(ctx->stream);
                prof_end(ctx);
                ++ctx->n_launches;
                ++n_issued;
                if (e != hipSuccess) rc = fail(ctx, "launch failed: %s", hipGetErrorString(e));
            }
            if (cur[i] < L.size()) any = true;
        }
        if (rc || !any) break;
        int lead = -1;
        for (int i = 0; i < n; ++i)
            if (cur[i] < ctx->rec[i].size()) { lead = i; break; }
        const RecOp& o0 = ctx->rec[lead][cur[lead]];
        const SplitArgs* list[tpz::SPLIT_MULTI_MAX];
        int who[tpz::SPLIT_MULTI_MAX], m = 0;
        double flops = 0, bytes = 0;
        for (int i = lead; i < n; ++i) {
            if (cur[i] >= ctx->rec[i].size()) continue;
            const RecOp& o = ctx->rec[i][cur[i]];
            if (o.ks != o0.ks || o.a.plan != o0.a.plan || split_mode_of(o.a) != split_mode_of(o0.a)) continue;
            // only what conv_split_multi_kernel is instantiated for merges (modes 0 / 1 / 2 / 11): a plane-stacked two-source
            // launch whose chunks mix both tensors (MODE 3: odd widths of a user-trained 3-D U-Net, TPZ_NO_SRCMAJOR) goes alone
            if (i != lead && (!o0.ks->launch_multi || split_mode_of(o0.a) == 3)) continue;
            list[m] = &o.a; who[m++] = i;
            flops += o.flops; bytes += o.bytes;
        }
        prof_begin(ctx, 0, flops, o0.ks->name, bytes);
        hipError_t e;
        if (m == 1) {
            SplitArgs a1 = o0.a;
            a1.n_tiles = 0;                    // (a launch of its own: n_tiles > 0 would select the persistent kernel)
            e = o0.ks->launch(a1, o0.grid, ctx->stream);
        } else {
            e = o0.ks->launch_multi(list, m, ctx->stream);
        }
        prof_end(ctx);
        ++ctx->n_launches;
        ++n_issued;
        if (e != hipSuccess) rc = fail(ctx, "conv_split launch failed: %s", hipGetErrorString(e));
        for (int k = 0; k < m; ++k) ++cur[who[k]];
    }
    if (g_trace_host) {
        size_t n_ops = 0;
        int n_img = 0;
        for (auto& r : ctx->rec) { n_ops += r.size(); n_img += r.empty() ? 0 : 1; }
        fprintf(stderr, "[tpz host] batch of %d images: %zu launches recorded in %.3f ms, issued as %lld in %.3f ms\n", n_img, n_ops,
                t_flush0 - ctx->rec_t0, n_issued, host_now_ms() - t_flush0);
    }
    for (auto& r : ctx->rec) r.clear();
    return rc;
}

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

static int upload(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out) {
    float* d = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&d, std::max<size_t>(n, 4) * sizeof(float)));
    HIPCHK(ctx, hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    if (m) m->dev_allocs.push_back(d);
    *out = d;
    return 0;
}

// per-channel vectors (bias, BN affine, head weights, weight scales) are zero-padded to whole 128-channel tiles plus one,
// so the epilogues fetch them as unclamped float4 loads; zero scale / bias make the padded channels come out as 0
static size_t chan_pad(size_t n) { return (n + 127) / 128 * 128 + 128; }
static int upload_chan(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out) {
    std::vector<float> padded(chan_pad(n), 0.f);
    memcpy(padded.data(), h, n * sizeof(float));
    return upload(ctx, m, padded.data(), padded.size(), out);
}

static const int MT_CHOICES[] = {16, 32, 48, 64, 96, 128};

// choose the MFMA instantiation for a conv layer; returns nullptr when the direct kernel must be used
static const ConvKernelInfo* pick_conv(int dims, int k, int dil, int cout, bool cin1, int epi) {
    const ConvKernelInfo* best = nullptr;
    int best_padded = 1 << 30;
    for (int mt : MT_CHOICES) {
        const ConvKernelInfo* c = find_conv(dims, k, dil, mt, cin1, epi);
        if (!c) continue;
        const int padded = (cout + mt - 1) / mt * mt;
        if (padded < best_padded || (padded == best_padded && best && mt > best->MT)) {
            best = c;
            best_padded = padded;
        }
    }
    return best;
}

static const ConvKernelInfo* choose_kernel(const tpz_layer& L) {
    if (L.cout == 1 && !L.head) return nullptr;      // M = 1: nothing for the matrix cores to do
    const bool cin1 = (L.cin == 1 && L.src2 < 0);
    // epilogue variant the layer needs (conv_mfma.h EPI_*)
    int epi = EPI_PLAIN;
    if (L.head) epi = EPI_HEAD;
    else if (L.res >= 0) epi = L.post_scale_off >= 0 ? EPI_RES_POST : EPI_RES;
    if ((L.head && (L.res >= 0 || L.post_scale_off >= 0)) || (L.res < 0 && L.post_scale_off >= 0)) return nullptr;
    return pick_conv(L.dims, L.k, L.dil, L.cout, cin1, epi);
}

// weights [cout][cin][kz][ky][kx] -> per (co-group, channel chunk, stage) blocks in A-fragment lane order:
//   block[step][mf][k(0..3)][i(0..15)]  with lane = k*16 + i   (conv_mfma.h)
//   generic: step = (kg*RPS + r)*K + kx, tap row = stage*RPS + r = kz*K + ky, ci = chunk*NCH + kg*4 + k
//   CIN1:    step = r*KXG + kxg,         kx = kxg*4 + k (zero beyond K), ci = 0
static void pack_weights(const ConvKernelInfo& ki, const float* w, int cout, int cin, int n_cog, int n_chunks,
                         std::vector<float>& out) {
    const int K = ki.K, KZ = ki.dims == 3 ? K : 1, MW = ki.MT / 16;
    out.assign((size_t)n_cog * n_chunks * ki.W_CHUNK, 0.f);
    const size_t taps = (size_t)KZ * K * K;
    for (int cog = 0; cog < n_cog; ++cog)
        for (int ch = 0; ch < n_chunks; ++ch)
            for (int j = 0; j < ki.SPG; ++j) {
                float* blk = out.data() + ((size_t)cog * n_chunks + ch) * ki.W_CHUNK + (size_t)j * ki.W_STAGE;
                for (int step = 0; step < ki.STEPS; ++step)
                    for (int mf = 0; mf < MW; ++mf)
                        for (int k = 0; k < 4; ++k)
                            for (int i = 0; i < 16; ++i) {
                                const int co = cog * ki.MT + mf * 16 + i;
                                int ci, kx, row;
                                if (ki.cin1) {
                                    row = j * ki.RPS + step / ki.KXG;
                                    kx = (step % ki.KXG) * 4 + k;
                                    ci = 0;
                                } else {
                                    const int kg = step / (ki.RPS * K);
                                    row = j * ki.RPS + (step / K) % ki.RPS;
                                    kx = step % K;
                                    ci = ch * ki.NCH + kg * 4 + k;
                                }
                                const int kz = row / K, ky = row % K;
                                float v = 0.f;
                                if (co < cout && ci < cin && kx < K)
                                    v = w[((size_t)co * cin + ci) * taps + ((size_t)kz * K + ky) * K + kx];
                                blk[((size_t)step * MW + mf) * 64 + k * 16 + i] = v;
                            }
            }
}

// Phase decomposition of conv(cat(upsample2x(a), b)) (the U-Net decoders, topaz/denoising/models.py:140-171).
// A k-tap "same" convolution of a 2x nearest-upsampled tensor touches only k1 = k/2 + 1 distinct source
// elements per axis; which ones, and with which sums of the original taps, depends on the parity p of the
// output coordinate:   source index = o + t - pad_p,   t(ky) = floor((p + ky - k/2) / 2) + pad_p,
// pad_p = -floor((p - k/2) / 2).  So the layer is run as 2^dims k1-tap convolutions over the LOW-resolution
// source `a` (one per output parity, weights pre-summed in fp64, output written to the strided positions)
// followed by the k-tap convolution of the skip source `b` alone, which adds itself in place and applies
// bias + activation.  The zero padding agrees because the upsample is exact (full = 2 * low per axis);
// run_conv() checks that at run time and otherwise keeps the fused upsample+concat loader.
static int phase_pad(int k, int p) { return (k / 2 - p + 1) / 2; }
static int phase_tap(int k, int p, int ky) {
    const int v = p + ky - k / 2;                  // floor(v / 2) for negative v too
    return (v >= 0 ? v / 2 : -((-v + 1) / 2)) + phase_pad(k, p);
}

static int prepare_phases(tpz_ctx* ctx, tpz_model* m, const tpz_layer& L, const float* w, int c1, int c2,
                          LayerRT& rt) {
    LayerRT::Phase& ph = rt.phase;
    if (L.src2 < 0 || L.dil != 1 || (L.k != 3 && L.k != 5) || L.pad != L.k / 2 || L.res >= 0 || L.head ||
        L.post_scale_off >= 0 || c1 + c2 != L.cin || c1 < 1 || c2 < 1)
        return 0;
    const int k = L.k, k1 = k / 2 + 1, dims = L.dims;
    ph.ki_low = pick_conv(dims, k1, 1, L.cout, false, EPI_PLAIN);
    ph.ki_skip = pick_conv(dims, k, 1, L.cout, c2 == 1, EPI_RES);
    if (!ph.ki_low || !ph.ki_skip) return 0;
    ph.c1 = c1; ph.c2 = c2; ph.k1 = k1;
    const int kz_n = dims == 3 ? k : 1, k1z_n = dims == 3 ? k1 : 1;
    const size_t taps = (size_t)kz_n * k * k, taps1 = (size_t)k1z_n * k1 * k1;
    ph.n_cog_low = (L.cout + ph.ki_low->MT - 1) / ph.ki_low->MT;
    ph.n_chunks_low = (c1 + ph.ki_low->NCH - 1) / ph.ki_low->NCH;
    ph.n_cog_skip = (L.cout + ph.ki_skip->MT - 1) / ph.ki_skip->MT;
    ph.n_chunks_skip = ph.ki_skip->cin1 ? 1 : (c2 + ph.ki_skip->NCH - 1) / ph.ki_skip->NCH;
    std::vector<double> acc;
    std::vector<float> eff, packed;
    const int n_phase = 1 << dims;
    for (int p = 0; p < n_phase; ++p) {
        const int px = p & 1, py = (p >> 1) & 1, pz = dims == 3 ? (p >> 2) & 1 : 0;
        acc.assign((size_t)L.cout * c1 * taps1, 0.0);
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < c1; ++ci)
                for (int kz = 0; kz < kz_n; ++kz)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx) {
                            const int tz = dims == 3 ? phase_tap(k, pz, kz) : 0;
                            const int ty = phase_tap(k, py, ky), tx = phase_tap(k, px, kx);
                            acc[((size_t)co * c1 + ci) * taps1 + ((size_t)tz * k1 + ty) * k1 + tx] +=
                                (double)w[((size_t)co * L.cin + ci) * taps + ((size_t)kz * k + ky) * k + kx];
                        }
        eff.resize(acc.size());
        for (size_t i = 0; i < acc.size(); ++i) eff[i] = (float)acc[i];
        pack_weights(*ph.ki_low, eff.data(), L.cout, c1, ph.n_cog_low, ph.n_chunks_low, packed);
        if (upload(ctx, m, packed.data(), packed.size(), &ph.d_w_low[p])) return 1;
    }
    eff.resize((size_t)L.cout * c2 * taps);
    for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < c2; ++ci)
            memcpy(&eff[((size_t)co * c2 + ci) * taps], &w[((size_t)co * L.cin + c1 + ci) * taps], taps * sizeof(float));
    pack_weights(*ph.ki_skip, eff.data(), L.cout, c2, ph.n_cog_skip, ph.n_chunks_skip, packed);
    if (upload(ctx, m, packed.data(), packed.size(), &ph.d_w_skip)) return 1;
    ph.valid = true;
    return 0;
}

static int prepare_layer(tpz_ctx* ctx, tpz_model* m, const tpz_layer& L, const float* blob, size_t n_floats,
                         LayerRT& rt, int c1 = 0, int c2 = 0) {
    rt.L = L;
    rt.c1 = c1; rt.c2 = c2;
    if (L.op != TPZ_OP_CONV) return 0;
    if (L.dims != 2 && L.dims != 3) return fail(ctx, "conv: dims must be 2 or 3");
    const size_t taps = (L.dims == 3 ? (size_t)L.k * L.k * L.k : (size_t)L.k * L.k);
    const size_t wn = (size_t)L.cout * L.cin * taps;
    if (L.w_off < 0 || (size_t)L.w_off + wn > n_floats) return fail(ctx, "conv: weight offset out of range");
    const float* w = blob + L.w_off;
    rt.ki = choose_kernel(L);
    if (rt.ki) {
        const ConvKernelInfo& ki = *rt.ki;
        rt.n_cog = (L.cout + ki.MT - 1) / ki.MT;
        rt.n_chunks = ki.cin1 ? 1 : (L.cin + ki.NCH - 1) / ki.NCH;
        rt.cog_inner = L.head ? rt.n_cog : 1;
        std::vector<float> packed;
        pack_weights(ki, w, L.cout, L.cin, rt.n_cog, rt.n_chunks, packed);
        if (upload(ctx, m, packed.data(), packed.size(), &rt.d_wpk)) return 1;
        if (!g_no_phase && prepare_phases(ctx, m, L, w, c1, c2, rt)) return 1;
    } else {
        if (L.src2 >= 0 || L.head || L.post_scale_off >= 0)
            return fail(ctx, "conv k=%d dil=%d cin=%d cout=%d dims=%d: no MFMA kernel compiled and the direct "
                        "kernel has no concat/head/affine epilogue", L.k, L.dil, L.cin, L.cout, L.dims);
        if (upload(ctx, m, w, wn, &rt.d_wpk)) return 1;
    }
    if (L.b_off >= 0) {
        if ((size_t)L.b_off + L.cout > n_floats) return fail(ctx, "conv: bias offset out of range");
        rt.bias0 = blob[L.b_off];
        if (upload_chan(ctx, m, blob + L.b_off, L.cout, &rt.d_bias)) return 1;
    }
    if (L.post_scale_off >= 0) {
        if (upload_chan(ctx, m, blob + L.post_scale_off, L.cout, &rt.d_post_scale)) return 1;
        if (upload_chan(ctx, m, blob + L.post_shift_off, L.cout, &rt.d_post_shift)) return 1;
    }
    if (L.head) {
        if (upload_chan(ctx, m, blob + L.head_w_off, L.cout, &rt.d_head_w)) return 1;
        rt.head_b = blob[L.head_b_off];
    }
    return 0;
}

// ---- 2xf16 path (conv_split.h) ---------------------------------------------------------------------
// weights [cout][cin][k][k] -> per (co-group, chunk, step) blocks  [plane hi|lo][m][lane = kb*16 + i][8 channels]
// of f16, scaled per output channel by 2^s (max |w| lands in [2^13, 2^14)) so that the lo halves stay normal.
// wp / cin_b: a 1x1 projection [cout][cin_b] folded in behind the conv's own stages, one step per chunk of its input cells
// (SplitArgs::fold_cells); mul: per-output-channel factor applied to both weight sets (an eval-BN scale folded into them)
static void pack_weights_split(const SplitKernelInfo& ki, const float* w, int cout, int cin, int n_cog, int n_chunks,
                               std::vector<uint16_t>& out, std::vector<float>& wscale_inv, const float* wp = nullptr,
                               int cin_b = 0, const float* mul = nullptr) {
    const int K = ki.K, KX = ki.KX, MW = ki.MT / 16;
    const size_t taps = (size_t)K * KX;
    std::vector<float> scale(cout, 1.f);
    wscale_inv.assign(cout, 1.f);
    for (int co = 0; co < cout; ++co) {
        float mx = 0.f;
        const float f = mul ? std::fabs(mul[co]) : 1.f;
        for (size_t i = 0; i < (size_t)cin * taps; ++i) mx = std::max(mx, f * std::fabs(w[(size_t)co * cin * taps + i]));
        for (int i = 0; wp && i < cin_b; ++i) mx = std::max(mx, f * std::fabs(wp[(size_t)co * cin_b + i]));
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) e = std::min(60, std::max(-60, (int)std::floor(std::log2(16384.0 / mx))));
        scale[co] = std::ldexp(1.f, e);
        wscale_inv[co] = std::ldexp(1.f, -e);
    }
    const size_t step_halfs = (size_t)ki.W_STEP_BYTES / 2;
    const int cells = (int)split_cells(cin);
    const int n_stages_a = ki.stages(cells);
    const int n_stages = n_stages_a + (wp ? (int)split_cells(cin_b) / ki.CC : 0);
    const int n_full = cells / ki.CC, n_rem = cells - n_full * ki.CC, taps_n = ki.cont ? ki.Q / ki.CC : 0;
    out.assign((size_t)n_cog * n_stages * step_halfs, 0);
    auto put = [&](uint16_t* blk, int m, int kb, int i, int j, float v) {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        uint16_t hb, lb;
        memcpy(&hb, &hi, 2);
        memcpy(&lb, &lo, 2);
        const size_t lane = (size_t)kb * 16 + i;
        blk[((size_t)(0 * MW + m) * 64 + lane) * 8 + j] = hb;
        blk[((size_t)(1 * MW + m) * 64 + lane) * 8 + j] = lb;
    };
    for (int cog = 0; cog < n_cog; ++cog)
        for (int st = 0; st < n_stages; ++st) {
            uint16_t* blk = out.data() + ((size_t)cog * n_stages + st) * step_halfs;
            if (st >= n_stages_a) {
                // folded projection: slots (centre tap, cell 0 .. CC-1); the slots of the neighbouring tap keep zero weights
                const int cb = st - n_stages_a;
                for (int kb = 0; kb < ki.CC; ++kb)
                    for (int m = 0; m < MW; ++m)
                        for (int i = 0; i < 16; ++i) {
                            const int co = cog * ki.MT + m * 16 + i;
                            if (co >= cout) continue;
                            for (int j = 0; j < 8; ++j) {
                                const int ci = (cb * ki.CC + kb) * 8 + j;
                                if (ci < cin_b) put(blk, m, kb, i, j, wp[(size_t)co * cin_b + ci] * scale[co] * (mul ? mul[co] : 1.f));
                            }
                        }
                continue;
            }
            for (int kb = 0; kb < 4; ++kb) {
                // (chunk, tap, cell) of lane group kb in this step
                int ch;
                SplitSlot sl;
                if (ki.cont) {
                    const int G = 4 * st + kb;
                    ch = G / ki.Q;
                    if (ch < n_full) {
                        sl = ki.cont_slot(G % ki.Q);
                    } else {
                        // short last chunk: (tap, cell) over its own n_rem cells; then the padding slots of the last step
                        const int q2 = G - n_full * ki.Q;
                        if (n_rem == 0 || q2 >= taps_n * n_rem) continue;
                        ch = n_full;
                        sl = ki.cont_slot((q2 / n_rem) * ki.CC + (q2 % n_rem));
                    }
                } else {
                    ch = st / ki.NSTEP;
                    sl = ki.slot(st % ki.NSTEP, kb);
                    if (sl.ky < 0) continue;
                }
                for (int m = 0; m < MW; ++m)
                    for (int i = 0; i < 16; ++i) {
                        const int co = cog * ki.MT + m * 16 + i;
                        if (co >= cout) continue;
                        for (int j = 0; j < 8; ++j) {
                            const int ci = (ch * ki.CC + sl.c) * 8 + j;
                            if (ci >= cin) continue;
                            put(blk, m, kb, i, j, w[((size_t)co * cin + ci) * taps + (size_t)sl.ky * KX + sl.kx] * scale[co] * (mul ? mul[co] : 1.f));
                        }
                    }
            }
        }
}

static const SplitKernelInfo* pick_split(int k, int dil, int cout, int epi, int kx = 0) {
    const SplitKernelInfo* best = nullptr;
    int best_padded = 1 << 30;
    for (int mt : MT_CHOICES) {
        const SplitKernelInfo* c = find_split(k, dil, mt, epi, kx);
        if (!c) continue;
        const int padded = (cout + mt - 1) / mt * mt;
        if (padded < best_padded || (padded == best_padded && best && mt > best->MT)) {
            best = c;
            best_padded = padded;
        }
    }
    return best;
}

static thread_local std::vector<uint16_t> g_pack_tmp;
static thread_local std::vector<float> g_inv_tmp;

// kz_n > 1: 3-D weights [cout][cin][kz][k][k] are laid out for the plane-stacked 2-D kernel (conv_split.h): the
// input channels of plane kz become channels [kz*cells*8, ...) of a 2-D conv with kz_n * cells * 8 input channels
// c1_major > 0 (a multiple of 8; two-source 3-D launches, SplitArgs::vol_srcmajor): the channels [0, c1_major) of every plane
// come first, then the remaining ones of every plane -- the order split_make_plan walks when srcmajor is set
static int upload_split_weights(tpz_ctx* ctx, tpz_model* m, const SplitKernelInfo& ks, const float* w, int cout, int cin,
                                int* n_cog, int* n_chunks, void** d_w, float** d_ws, int kz_n = 1, int c1_major = 0) {
    std::vector<float> stacked;
    if (kz_n > 1) {
        const int c8 = (int)split_cells(cin) * 8, k = ks.K;
        const size_t taps2 = (size_t)k * ks.KX;
        stacked.assign((size_t)cout * kz_n * c8 * taps2, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int kz = 0; kz < kz_n; ++kz) {
                    size_t vch = (size_t)kz * c8 + ci;                                  // plane-major
                    if (c1_major > 0)
                        vch = ci < c1_major ? (size_t)kz * c1_major + ci
                                            : (size_t)kz_n * c1_major + (size_t)kz * (c8 - c1_major) + (ci - c1_major);
                    memcpy(&stacked[((size_t)co * kz_n * c8 + vch) * taps2],
                           &w[(((size_t)co * cin + ci) * kz_n + kz) * taps2], taps2 * sizeof(float));
                }
        w = stacked.data();
        cin = kz_n * c8;
    }
    *n_cog = (cout + ks.MT - 1) / ks.MT;
    *n_chunks = (int)((split_cells(cin) + ks.CC - 1) / ks.CC);
    std::vector<uint16_t> packed;
    std::vector<float> inv;
    pack_weights_split(ks, w, cout, cin, *n_cog, *n_chunks, packed, inv);
    if (!d_w) {                                  // caller concatenates: hand the host vectors back
        g_pack_tmp.swap(packed);
        g_inv_tmp.swap(inv);
        return 0;
    }
    float* d = nullptr;
    if (upload(ctx, m, reinterpret_cast<const float*>(packed.data()), (packed.size() + 1) / 2, &d)) return 1;
    *d_w = d;
    return upload_chan(ctx, m, inv.data(), inv.size(), d_ws);
}

// 2xf16 twin of prepare_phases for a 2-D decoder layer conv(cat(upsample2x(a), b)); needs rt.phase (fp32)
static int prepare_split_phases(tpz_ctx* ctx, tpz_model* m, const float* w, LayerRT& rt) {
    const tpz_layer& L = rt.L;
    const LayerRT::Phase& ph = rt.phase;
    LayerRT::SplitPhase& sp = rt.sphase;
    if (!ph.valid || ph.c1 % 8 != 0) return 0;
    const int k = L.k, k1 = ph.k1, c1 = ph.c1, c2 = ph.c2, dims = L.dims;
    const int kz_n = dims == 3 ? k : 1, k1z_n = dims == 3 ? k1 : 1;
    sp.ks_low = pick_split(k1, 1, L.cout, EPI_RES);
    if (!sp.ks_low) return 0;
    if (c2 == 1) {
        // same tile and weight packing as the fp32 skip kernel of prepare_phases: its packed weights are reused
        if (!ph.ki_skip->cin1) return 0;
        sp.ki_skip_stem = find_conv(dims, k, 1, ph.ki_skip->MT, true, EPI_SPLIT);
        if (!sp.ki_skip_stem || ph.n_cog_skip != 1) return 0;
    } else {
        sp.ks_skip = pick_split(k, 1, L.cout, EPI_PLAIN);
        if (!sp.ks_skip) return 0;
    }
    const size_t taps = (size_t)kz_n * k * k, taps1 = (size_t)k1z_n * k1 * k1;
    // 3x3(x3) with a 1-channel skip source: its space-to-depth cell reads exactly the 2-tap window of each parity, so it
    // joins every parity kernel as one more input cell (8 channels) and the skip pass + in-place residual disappear
    if (k == 3 && c2 == 1) {
        sp.ks_low_plain = find_split(k1, 1, sp.ks_low->MT, EPI_PLAIN);
        // (2-D: the chunks switch source, so the first source must fill whole chunks; 3-D picks the source per cell)
        if (sp.ks_low_plain && sp.ks_low_plain->CC == sp.ks_low->CC && sp.ks_low_plain->WAVES == sp.ks_low->WAVES &&
            (dims == 3 || (c1 / 8) % sp.ks_low_plain->CC == 0))
            sp.low_with_skip = true;
    }
    const int c1e = sp.low_with_skip ? c1 + 8 : c1;
    std::vector<double> acc;
    std::vector<float> eff, all_s, sub_w;
    std::vector<uint16_t> all_w;
    for (int p = 0; p < (1 << dims); ++p) {
        const int px = p & 1, py = (p >> 1) & 1, pz = dims == 3 ? (p >> 2) & 1 : 0;
        acc.assign((size_t)L.cout * c1 * taps1, 0.0);
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < c1; ++ci)
                for (int kz = 0; kz < kz_n; ++kz)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx) {
                            const int tz = dims == 3 ? phase_tap(k, pz, kz) : 0;
                            acc[((size_t)co * c1 + ci) * taps1 + ((size_t)tz * k1 + phase_tap(k, py, ky)) * k1 + phase_tap(k, px, kx)] +=
                                (double)w[((size_t)co * L.cin + ci) * taps + ((size_t)kz * k + ky) * k + kx];
                        }
        eff.assign((size_t)L.cout * c1e * taps1, 0.f);
        for (int co = 0; co < L.cout; ++co)
            for (size_t i = 0; i < (size_t)c1 * taps1; ++i) eff[(size_t)co * c1e * taps1 + i] = (float)acc[(size_t)co * c1 * taps1 + i];
        if (sp.low_with_skip) {
            const int nq = 1 << dims;
            for (int co = 0; co < L.cout; ++co)
                for (int q = 0; q < nq; ++q) {
                    const int qx = q & 1, qy = (q >> 1) & 1, qz = dims == 3 ? (q >> 2) & 1 : 0;
                    for (int tz = 0; tz < k1z_n; ++tz)
                        for (int ty = 0; ty < k1; ++ty)
                            for (int tx = 0; tx < k1; ++tx) {
                                // full-resolution offset of s2d element (q, tap t) from the output voxel of parity p
                                const int dz = dims == 3 ? 2 * (tz - phase_pad(k, pz)) + qz - pz : 0;
                                const int dy = 2 * (ty - phase_pad(k, py)) + qy - py, dx = 2 * (tx - phase_pad(k, px)) + qx - px;
                                if (dz < -1 || dz > 1 || dy < -1 || dy > 1 || dx < -1 || dx > 1) continue;
                                const int kz = dims == 3 ? dz + 1 : 0;
                                eff[((size_t)co * c1e + c1 + q) * taps1 + ((size_t)tz * k1 + ty) * k1 + tx] =
                                    w[((size_t)co * L.cin + c1) * taps + ((size_t)kz * k + (dy + 1)) * k + (dx + 1)];
                            }
                }
        }
        if (!sp.low_with_skip) sub_w.insert(sub_w.end(), eff.begin(), eff.end());          // [parity][cout][c1][taps1]
        // 3-D with the skip cell: source-major cell order whenever the first source's planes fill whole chunks (then no chunk of
        // the K loop mixes the two tensors: conv_split.h MODE 11)
        const bool no_srcmajor = getenv("TPZ_NO_SRCMAJOR") != nullptr;   // A/B switch, read when the model is loaded
        sp.srcmajor = dims == 3 && sp.low_with_skip && !no_srcmajor && (k1z_n * (c1 / 8)) % sp.ks_low_plain->CC == 0;
        if (upload_split_weights(ctx, m, sp.low_with_skip ? *sp.ks_low_plain : *sp.ks_low, eff.data(), L.cout, c1e,
                                 &sp.n_cog_low, &sp.n_chunks_low, nullptr, nullptr, k1z_n, sp.srcmajor ? c1 : 0)) return 1;
        sp.w_phase_bytes = g_pack_tmp.size() * sizeof(uint16_t);
        all_w.insert(all_w.end(), g_pack_tmp.begin(), g_pack_tmp.end());
        g_inv_tmp.resize(chan_pad(L.cout), 0.f);                      // stride chan_pad(cout) per parity
        all_s.insert(all_s.end(), g_inv_tmp.begin(), g_inv_tmp.end());
    }
    // 5x5 (2-D): both parities of an axis read the same 3-tap window, so the four parity kernels share their B
    // operand: one conv with 4*cout virtual output channels on the 128-channel tile (conv_split.h subpix_cout)
    if (dims == 2 && k == 5 && L.cout % 16 == 0) sp.ks_sub = find_split(k1, 1, 128, EPI_RES);
    if (sp.ks_sub && c2 == 1) {
        // The 1-channel skip source x joins as 4 space-to-depth channels (s2d_split_kernel): x[2y+qy][2x+qx] is channel
        // 2*qy+qx of the low-resolution pixel (y, x), and the 5x5 window around output (2oy+py, 2ox+px) lies inside the
        // same 3x3 low-resolution window: tap (ty, tx) of s2d channel (qy, qx) carries w[ky][kx], ky = 2(ty-1)+qy-py+2.
        // One plain launch then does the whole layer -- no skip pass, no in-place residual.
        const SplitKernelInfo* pl = find_split(k1, 1, 128, EPI_PLAIN);
        if (pl && pl->CC == sp.ks_sub->CC) {
            const int cin2 = c1 + 8;
            std::vector<float> w2((size_t)4 * L.cout * cin2 * taps1, 0.f);
            for (int p = 0; p < 4; ++p)
                for (int co = 0; co < L.cout; ++co) {
                    const size_t v = (size_t)p * L.cout + co;
                    memcpy(&w2[v * cin2 * taps1], &sub_w[v * c1 * taps1], (size_t)c1 * taps1 * sizeof(float));
                    const int px = p & 1, py = (p >> 1) & 1;
                    for (int q = 0; q < 4; ++q)
                        for (int ty = 0; ty < 3; ++ty)
                            for (int tx = 0; tx < 3; ++tx) {
                                const int ky = 2 * (ty - 1) + (q >> 1) - py + 2, kx = 2 * (tx - 1) + (q & 1) - px + 2;
                                if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
                                w2[(v * cin2 + c1 + q) * taps1 + (size_t)ty * 3 + tx] = w[((size_t)co * L.cin + c1) * taps + (size_t)ky * 5 + kx];
                            }
                }
            sub_w.swap(w2);
            sp.ks_sub = pl;
            sp.sub_with_skip = true;
        }
    }
    if (sp.ks_sub) {
        all_w.clear(); all_s.clear();
        int nch = 0;
        void* dw = nullptr;
        if (upload_split_weights(ctx, m, *sp.ks_sub, sub_w.data(), 4 * L.cout, sp.sub_with_skip ? c1 + 8 : c1, &sp.n_cog_sub, &nch, &dw, &sp.d_ws_low)) return 1;
        sp.d_w_low = dw;
        sp.n_chunks_low = nch;
    } else {
        float* d = nullptr;
        if (upload(ctx, m, reinterpret_cast<const float*>(all_w.data()), (all_w.size() + 1) / 2, &d)) return 1;
        sp.d_w_low = d;
        if (upload(ctx, m, all_s.data(), all_s.size(), &sp.d_ws_low)) return 1;
    }
    if (sp.ks_skip) {
        eff.resize((size_t)L.cout * c2 * taps);
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < c2; ++ci)
                memcpy(&eff[((size_t)co * c2 + ci) * taps], &w[((size_t)co * L.cin + c1 + ci) * taps], taps * sizeof(float));
        if (upload_split_weights(ctx, m, *sp.ks_skip, eff.data(), L.cout, c2, &sp.n_cog_skip, &sp.n_chunks_skip,
                                 &sp.d_w_skip, &sp.d_ws_skip, kz_n)) return 1;
    }
    sp.valid = true;
    return 0;
}

// Chooses, layer by layer, what can run on the 2xf16 path (2-D programs only):
//   * single-source convs with a conv_split kernel for their (k, dilation, cout, epilogue);
//   * decoder convs over an upsampled + a skip source through the per-parity twin (prepare_split_phases);
//   * 1-channel stems keep their fp32 MFMA kernel but store split cells when their consumers read them;
//   * max-pooling runs in whichever format its source has.
// A conv_split layer whose consumers all read fp32 (the 1-output-channel last conv of the U-Nets runs on the
// direct kernel) takes the fp32-storing variant when one is compiled; everything else that meets a tensor in the
// other format has it converted on the device (run_program / slot_as).
static int prepare_split(tpz_ctx* ctx, tpz_model* m, const float* blob) {
    const int nl = (int)m->layers.size();
    // does layer j read slot `slot` as split cells?  (max-pool: whatever its own consumers read)
    std::vector<int> reads(nl, 0);              // per conv layer: 1 = its (non-image) sources are read as split
    for (int i = 0; i < nl; ++i) {
        LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        // the split epilogue applies the activation as max(v, slope * v): right for every slope <= 1 (ReLU, LeakyReLU,
        // identity, PReLU as trained); a layer with a larger slope stays on its fp32 kernel
        if (L.op == TPZ_OP_CONV && L.slope > 1.f) continue;
        if (L.op == TPZ_OP_CONV && !rt.ki && L.cout == 1 && L.cin % 8 == 0 && L.src2 < 0 && (L.res < 0 || L.res_crop == 0) && !L.head &&
            L.post_scale_off < 0 && L.dil == 1 && L.pad == L.k / 2 && L.slope == 1.f && i == nl - 1) {
            // 1-output-channel last conv: its kx taps as k virtual output channels of a k x 1 column kernel
            rt.ks_last = find_split(L.k, 1, 16, EPI_PLAIN_F32, 1);
            reads[i] = rt.ks_last ? 1 : 0;
            continue;
        }
        if (L.op != TPZ_OP_CONV || !rt.ki || rt.ki->cin1) continue;
        if (L.src2 >= 0) {
            if (prepare_split_phases(ctx, m, blob + L.w_off, rt)) return 1;
            // any other geometry (2-D): the same kernel family with the upsample + concat folded into its loader
            if (L.dims == 2 && rt.ki->epi == EPI_PLAIN && rt.c1 + rt.c2 == L.cin) {
                rt.ks = pick_split(L.k, L.dil, L.cout, EPI_PLAIN);
                if (rt.ks && rt.c1 % (8 * rt.ks->CC) != 0) rt.ks = nullptr;
            }
            reads[i] = (rt.sphase.valid || rt.ks) ? 1 : 0;
            continue;
        }
        rt.ks = pick_split(L.k, L.dil, L.cout, rt.ki->epi);
        if (!rt.ks && rt.ki->epi == EPI_PLAIN) rt.ks = pick_split(L.k, L.dil, L.cout, EPI_PLAIN_F32);
        reads[i] = rt.ks ? 1 : 0;
    }
    std::function<bool(int)> slot_read_split = [&](int slot) {
        bool any = false;
        for (int j = 0; j < nl; ++j) {
            const tpz_layer& Lj = m->layers[j].L;
            const bool uses = Lj.src == slot || Lj.src2 == slot || Lj.res == slot;
            if (!uses) continue;
            if (Lj.op == TPZ_OP_MAXPOOL2 || Lj.op == TPZ_OP_MAXPOOL) any |= slot_read_split(Lj.dst);   // pools keep the format
            else if (Lj.src2 == slot && m->layers[j].sphase.valid && m->layers[j].sphase.ki_skip_stem) continue;  // fp32
            else any |= reads[j] != 0;
        }
        return any;
    };
    bool any_split = false;
    for (int i = 0; i < nl; ++i) {
        LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        if (L.op != TPZ_OP_CONV) continue;
        const bool wanted = slot_read_split(L.dst);
        if (rt.ki && rt.ki->cin1 && L.src2 < 0) {
            if (!(wanted && L.res < 0 && !L.head && L.post_scale_off < 0)) continue;
            // stem: a k x 1 column kernel over an x-shifted copy of the image (kx taps as 8*ncell input channels) ...
            if (L.dil == 1 && !(L.slope > 1.f)) rt.ks_stem = pick_split(L.k, 1, L.cout, EPI_PLAIN, 1);
            if (rt.ks_stem) {
                const int k = L.k, kz_n = L.dims == 3 ? k : 1, c8 = (k + 7) / 8 * 8;
                std::vector<float> w2((size_t)L.cout * c8 * kz_n * k, 0.f);
                const float* w = blob + L.w_off;                     // [cout][1][kz][ky][kx]
                for (int co = 0; co < L.cout; ++co)
                    for (int kz = 0; kz < kz_n; ++kz)
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx)
                                w2[(((size_t)co * c8 + kx) * kz_n + kz) * k + ky] = w[(((size_t)co * kz_n + kz) * k + ky) * k + kx];
                if (upload_split_weights(ctx, m, *rt.ks_stem, w2.data(), L.cout, c8, &rt.s_n_cog, &rt.s_n_chunks,
                                         &rt.d_wsplit, &rt.d_wscale, kz_n)) return 1;
                any_split = true;
            } else if (rt.n_cog == 1) {
                // ... or the fp32 MFMA kernel with a split store
                rt.ki_stem_split = find_conv(L.dims, L.k, L.dil, rt.ki->MT, true, EPI_SPLIT);
            }
            continue;
        }
        if (rt.ks_last) {
            const int k = L.k, kz_n = L.dims == 3 ? k : 1;
            const float* w = blob + L.w_off;                         // [1][cin][kz][ky][kx]
            // k = 3 / 5 over <= 1024 taps x channels (Conv(32, 1, 5), Conv3d(32, 1, 3)): the fp32 stencil on the vector ALUs
            // (kernels_misc.hip conv_cout1_split_kernel) instead of a 16-row MFMA tile with one useful row per kx tap
            static const bool no_valu_last = getenv("TPZ_NO_VALU_LAST") != nullptr;       // A/B switch
            if (!no_valu_last && (k == 3 || k == 5) && (size_t)L.cin * k * k * kz_n <= 1024) {
                const int cells = (int)split_cells(L.cin);
                std::vector<float> wl((size_t)kz_n * cells * k * k * 8, 0.f);
                for (int ci = 0; ci < L.cin; ++ci)
                    for (int kz = 0; kz < kz_n; ++kz)
                        for (int ky = 0; ky < k; ++ky)
                            for (int kx = 0; kx < k; ++kx)
                                wl[((((size_t)kz * cells + ci / 8) * k + kx) * k + ky) * 8 + ci % 8] =
                                    w[(((size_t)ci * kz_n + kz) * k + ky) * k + kx];
                if (upload(ctx, m, wl.data(), wl.size(), &rt.d_wlast)) return 1;
            }
            std::vector<float> w2((size_t)k * L.cin * kz_n * k);
            for (int v = 0; v < k; ++v)
                for (int ci = 0; ci < L.cin; ++ci)
                    for (int kz = 0; kz < kz_n; ++kz)
                        for (int ky = 0; ky < k; ++ky)
                            w2[(((size_t)v * L.cin + ci) * kz_n + kz) * k + ky] = w[(((size_t)ci * kz_n + kz) * k + ky) * k + v];
            if (upload_split_weights(ctx, m, *rt.ks_last, w2.data(), k, L.cin, &rt.s_n_cog, &rt.s_n_chunks,
                                     &rt.d_wsplit, &rt.d_wscale, kz_n)) return 1;
            any_split = true;
            continue;
        }
        if (rt.ks) {
            if (!L.head && !wanted && rt.ks->epi == EPI_PLAIN && L.src2 < 0) {
                const SplitKernelInfo* f = find_split(L.k, L.dil, rt.ks->MT, EPI_PLAIN_F32);
                if (!f) f = pick_split(L.k, L.dil, L.cout, EPI_PLAIN_F32);
                if (f) rt.ks = f;
            }
            if (i == nl - 1 && !L.head && rt.ks->epi != EPI_PLAIN_F32) rt.ks = nullptr;    // the result leaves as fp32
        }
        if (rt.ks) {
            if (upload_split_weights(ctx, m, *rt.ks, blob + L.w_off, L.cout, L.cin, &rt.s_n_cog, &rt.s_n_chunks,
                                     &rt.d_wsplit, &rt.d_wscale, L.dims == 3 ? L.k : 1)) return 1;
            any_split = true;
            // the weights-resident kernel for the 3x3 32 -> 32 layers (conv_rw.h): same tensors either side, its own weight order
            static const bool no_rw = getenv("TPZ_NO_RW") != nullptr;                      // A/B switch
            if (!no_rw && L.dims == 2 && L.k == 3 && L.cin == 32 && L.cout == 32 && L.src2 < 0 && !L.head &&
                (L.dil == 1 || L.dil == 2 || L.dil == 4) && rt.ks->epi <= EPI_RES_POST) {
                SplitKernelInfo rw;
                memset(&rw, 0, sizeof rw);
                rw.K = rw.KX = 3; rw.D = L.dil; rw.MT = 32; rw.CC = 4; rw.cont = 1; rw.Q = 36; rw.SPS = 1;
                rw.W_STEP_BYTES = 2 * (32 / 16) * 1024;
                rw.cont_slot = [](int q) { return SplitSlot{(q / 4) / 3, (q / 4) % 3, q % 4}; };
                int n_cog = 0, n_chunks = 0;
                if (upload_split_weights(ctx, m, rw, blob + L.w_off, L.cout, L.cin, &n_cog, &n_chunks, &rt.d_w_rw, &rt.d_ws_rw)) return 1;
            }
        }
        if (rt.sphase.valid) any_split = true;
    }
    // conv -> MaxPool2d(2) where nothing else reads the conv's output (the U-Net encoders): pool in the conv's epilogue
    for (int i = 0; i + 1 < nl; ++i) {
        LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        // (3-D: the plane-stacked kernels pool in-plane, maxpoolz_split_kernel finishes the z pairs)
        static const bool no_pool3d = getenv("TPZ_NO_POOL3D") != nullptr;       // A/B switch
        if (L.op != TPZ_OP_CONV || L.dil != 1 || L.head || L.res >= 0 || L.post_scale_off >= 0 || (L.dims == 3 && no_pool3d)) continue;
        const SplitKernelInfo* base = rt.ks_stem ? rt.ks_stem : ((rt.ks && L.src2 < 0 && rt.ks->epi == EPI_PLAIN) ? rt.ks : nullptr);
        if (!base) continue;
        int readers = 0, pool = -1;
        for (int j = 0; j < nl; ++j) {
            const tpz_layer& Lj = m->layers[j].L;
            if (Lj.src == L.dst || Lj.src2 == L.dst || Lj.res == L.dst) { ++readers; if (Lj.op == TPZ_OP_MAXPOOL2 && Lj.src == L.dst) pool = j; }
        }
        if (readers != 1 || pool != i + 1) continue;
        const SplitKernelInfo* pk = find_split(base->K, base->D, base->MT, EPI_POOL, base->KX);
        if (pk && pk->CC == base->CC && pk->NSTEP == base->NSTEP && pk->cont == base->cont && pk->W_STEP_BYTES == base->W_STEP_BYTES) rt.ks_pool = pk;
    }
    // ---- fold 1x1 projections into the conv that adds them as its residual
    static const bool no_fold = getenv("TPZ_NO_FOLD") != nullptr;
    for (int i = 0; i < nl && !no_fold; ++i) {
        LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        if (L.op != TPZ_OP_CONV || L.dims != 2 || !rt.ks || L.res < 0 || L.src2 >= 0 || L.head || L.k % 2 == 0) continue;
        if (rt.ks->epi != EPI_RES && rt.ks->epi != EPI_RES_POST) continue;
        int j = -1, readers = 0;
        for (int t = 0; t < nl; ++t) {
            const tpz_layer& T = m->layers[t].L;
            if (T.dst == L.res && t < i) j = t;
            if (T.src == L.res || T.src2 == L.res || T.res == L.res) ++readers;
        }
        if (j < 0 || readers != 1) continue;
        LayerRT& pj = m->layers[j];
        const tpz_layer& P = pj.L;
        if (P.op != TPZ_OP_CONV || P.dims != 2 || P.k != 1 || P.pad != 0 || P.slope != 1.f || P.b_off >= 0 || P.res >= 0 ||
            P.src2 >= 0 || P.head || P.post_scale_off >= 0 || P.cout != L.cout || !pj.ks) continue;
        const SplitKernelInfo* kf = find_split(L.k, L.dil, rt.ks->MT, EPI_PLAIN, 0, 1);
        if (!kf || !kf->cont || kf->CC != 2 || L.cin % 16 != 0 || P.cin % 16 != 0) continue;
        // the slot the projection reads must hold split cells when conv1 runs: it does if a 2xf16 layer reads it anyway
        std::vector<float> mul, bias(L.cout, 0.f);
        if (L.b_off >= 0) memcpy(bias.data(), blob + L.b_off, L.cout * sizeof(float));
        if (L.post_scale_off >= 0) {
            mul.assign(blob + L.post_scale_off, blob + L.post_scale_off + L.cout);
            for (int c = 0; c < L.cout; ++c) bias[c] = bias[c] * mul[c] + blob[L.post_shift_off + c];
        }
        rt.f_n_cog = (L.cout + kf->MT - 1) / kf->MT;
        rt.fold_cells = (int)split_cells(P.cin);
        rt.f_n_chunks = (int)split_cells(L.cin) / kf->CC + rt.fold_cells / kf->CC;
        std::vector<uint16_t> packed;
        std::vector<float> inv;
        pack_weights_split(*kf, blob + L.w_off, L.cout, L.cin, rt.f_n_cog, rt.f_n_chunks, packed, inv, blob + P.w_off, P.cin,
                           mul.empty() ? nullptr : mul.data());
        float* d = nullptr;
        if (upload(ctx, m, reinterpret_cast<const float*>(packed.data()), (packed.size() + 1) / 2, &d)) return 1;
        rt.d_wfold = d;
        if (upload_chan(ctx, m, inv.data(), inv.size(), &rt.d_wscale_fold)) return 1;
        if (upload_chan(ctx, m, bias.data(), bias.size(), &rt.d_bias_fold)) return 1;
        rt.ks_fold = kf;
        rt.fold_src = P.src;
        pj.folded_into = i;
        m->last_use[P.src] = std::max(m->last_use[P.src], i);       // conv1 now reads the projection's input itself
    }
    m->split_ok = any_split;
    // how much of the model the 2xf16 path covers (tpz_model_split_layers): a mixed program is correct -- the other layers run
    // on their fp32 kernels with a format conversion either side -- but several times slower than it looks
    m->n_conv = m->n_conv_split = 0;
    m->off_path.clear();
    for (int i = 0; i < nl; ++i) {
        const LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        if (L.op != TPZ_OP_CONV) continue;
        ++m->n_conv;
        const bool on = rt.ks || rt.ks_stem || rt.ks_last || rt.sphase.valid || rt.ki_stem_split ||
                        (rt.folded_into >= 0 && m->layers[rt.folded_into].ks_fold);
        if (on) { ++m->n_conv_split; continue; }
        char buf[96];
        snprintf(buf, sizeof buf, "%s#%d %dx%d d%d %d->%d", m->off_path.empty() ? "" : ", ", i, L.k, L.k, L.dil, L.cin, L.cout);
        if (m->off_path.size() < 400) m->off_path += buf;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// executor
// ------------------------------------------------------------------------------------------------
// grid, XCD swizzle and phase stagger of one conv_mfma launch; a.Dout/Hout/Wout, n_chunks, cog_inner are set
static int launch_mfma(tpz_ctx* ctx, const ConvKernelInfo& ki, ConvArgs& a, int n_cog, double flops) {
    a.xcd_swizzle = 1;
    if (a.wy1 <= 0) { a.wy0 = a.wx0 = 0; a.wy1 = a.Hout; a.wx1 = a.Wout; }      // no window: the whole lattice
    else flops *= (double)(a.wy1 - a.wy0) * (a.wx1 - a.wx0) / ((double)a.Hout * a.Wout);
    if (a.wz1 <= 0 || ki.dims != 3) { a.wz0 = 0; a.wz1 = std::max(a.Dout, 1); }  // no z window (every 2-D launch): the whole depth
    else flops *= (double)(a.wz1 - a.wz0) / a.Dout;
    a.tiles_x = (a.wx1 - a.wx0 + ki.TW - 1) / ki.TW;
    a.tiles_y = (a.wy1 - a.wy0 + ki.TH * ki.D - 1) / (ki.TH * ki.D) * ki.D;
    a.tiles_z = ki.dims == 3 ? (a.wz1 - a.wz0 + ki.TD * ki.D - 1) / (ki.TD * ki.D) * ki.D : 1;
    a.stagger_first = a.stagger_sleeps = 0;
    // phase stagger of the two workgroups per CU (conv_mfma.h); only worth it for many generations
    if ((long long)a.tiles_x * a.tiles_y >= 4096) {
        a.stagger_first = 512;
        a.stagger_sleeps = (int)((long long)a.n_chunks * ki.SPG * ki.STEPS * (ki.MT / 16) *
                                 ((ki.TD * ki.TH / 4) * (ki.TW / 16)) * 32 / 8128 / 2);
    }
    if ((long long)a.tiles_y * a.tiles_z > 65535) return fail(ctx, "conv grid too large");
    // 32-bit LDS-DMA byte offsets relative to the first channel of a chunk
    if ((size_t)ki.NCH * (size_t)std::max(a.cs1, a.cs2) * 4 >= ((size_t)1 << 32))
        return fail(ctx, "image too large for one launch: process it in patches");
    dim3 grid(a.tiles_x, a.tiles_y * a.tiles_z, n_cog / a.cog_inner);
    const ConvKernelInfo* kip = &ki;
    const ConvArgs ac = a;
    hipError_t e = enqueue(ctx, 0, flops, ki.name, 0.0, [kip, ac, grid](hipStream_t st) { return kip->launch(ac, grid, st); });
    HIPCHK(ctx, e);
    return 0;
}

// launch window of an fp32 kernel from the part of the layer's tensor that is needed (`scale` = 2: the half-resolution lattice of
// a per-parity launch).  The left edge is rounded down to a multiple of 4 pixels: the 16-byte granules of the MFMA kernels'
// loader stay aligned; the few extra columns are computed like any others.
static void set_window(ConvArgs& a, const Rect& need, int scale = 1) {
    if (!need.on) return;
    a.wy0 = need.y0 / scale; a.wx0 = (need.x0 / scale) & ~3;
    a.wy1 = std::min(a.Hout, (need.y1 + scale - 1) / scale);
    a.wx1 = std::min(a.Wout, (need.x1 + scale - 1) / scale);
    a.wy1 = std::max(a.wy1, a.wy0 + 1); a.wx1 = std::max(a.wx1, a.wx0 + 1);
    if (a.Dout > 1) {          // 3-D: the planes of the box
        a.wz0 = std::min(a.Dout - 1, need.z0 / scale);
        a.wz1 = std::max(a.wz0 + 1, std::min(a.Dout, (need.z1 + scale - 1) / scale));
    }
}

// conv(cat(upsample2x(s1), s2)) by output parity (prepare_phases): 2^dims plain launches over s1 that write the
// strided output positions, then the skip-source launch over the full grid that adds itself in place.
static int run_conv_phases(tpz_ctx* ctx, const LayerRT& rt, const ConvArgs& base, const Slot& s1, const Slot& s2,
                           Slot& dst) {
    const tpz_layer& L = rt.L;
    const LayerRT::Phase& ph = rt.phase;
    const int n_phase = 1 << L.dims;
    for (int p = 0; p < n_phase; ++p) {
        const int px = p & 1, py = (p >> 1) & 1, pz = L.dims == 3 ? (p >> 2) & 1 : 0;
        ConvArgs a = base;
        a.in2 = nullptr;
        a.wpk = ph.d_w_low[p];
        a.bias = nullptr;
        a.res = nullptr;
        a.nrm = nullptr;
        a.norm_out = 0;
        a.slope = 1.f;
        a.Cin = a.Cin1 = ph.c1;
        a.Din = a.D1 = s1.D; a.Hin = a.H1 = s1.H; a.Win = a.W1 = s1.W;
        a.Dout = s1.D; a.Hout = s1.H; a.Wout = s1.W;                  // the lattice of this parity
        a.pad_x = phase_pad(L.k, px); a.pad_y = phase_pad(L.k, py); a.pad_z = L.dims == 3 ? phase_pad(L.k, pz) : 0;
        a.pad = a.pad_x;
        a.os = 2; a.oox = px; a.ooy = py; a.ooz = pz;
        a.n_chunks = ph.n_chunks_low;
        a.cog_inner = 1;
        a.wy0 = a.wx0 = a.wy1 = a.wx1 = a.wz0 = a.wz1 = 0;
        set_window(a, dst.need, 2);
        const double fl = 2.0 * L.cout * ph.c1 * std::pow((double)ph.k1, L.dims) * (double)s1.D * s1.H * s1.W;
        if (launch_mfma(ctx, *ph.ki_low, a, ph.n_cog_low, fl)) return 1;
    }
    ConvArgs a = base;
    a.in = s2.p;
    a.in2 = nullptr;
    a.wpk = ph.d_w_skip;
    a.res = dst.p;                                                     // in place: every thread reads what it writes
    a.Dres = dst.D; a.Hres = dst.H; a.Wres = dst.W; a.res_crop = 0;
    a.Cin = a.Cin1 = ph.c2;
    a.D1 = s2.D; a.H1 = s2.H; a.W1 = s2.W;
    a.cs1 = s2.cs; a.ps1 = s2.ps; a.pitch1 = s2.pitch;
    a.n_chunks = ph.n_chunks_skip;
    a.cog_inner = 1;
    const double fl = 2.0 * L.cout * ph.c2 * std::pow((double)L.k, L.dims) * (double)dst.D * dst.H * dst.W;
    return launch_mfma(ctx, *ph.ki_skip, a, ph.n_cog_skip, fl);
}

static int launch_split(tpz_ctx* ctx, const SplitKernelInfo& ks, SplitArgs& a, int n_cog, double flops);

// window of a launch from the part of the layer's tensor that is needed (`need` in the tensor's coordinates, `scale` = 2 for the
// low-resolution lattice of a per-parity / sub-pixel launch, `grow_x` extra columns at the right: the column kernel of a last
// conv); flops are scaled by the fraction of the lattice that is computed
static void set_window(SplitArgs& a, const Rect& need, int scale = 1, int grow_x = 0) {
    if (!need.on) return;
    a.wy0 = need.y0 / scale; a.wx0 = need.x0 / scale;
    a.wy1 = std::min(a.Hout, (need.y1 + scale - 1) / scale);
    a.wx1 = std::min(a.Wout, (need.x1 + scale - 1) / scale + grow_x);
    a.wy1 = std::max(a.wy1, a.wy0 + 1); a.wx1 = std::max(a.wx1, a.wx0 + 1);
    a.wy1 = -a.wy1;            // (marks the window as set: launch_split flips it back)
    if (a.Dout > 1) {          // plane-stacked 3-D: the planes of the box
        a.Dlat = a.Dout;
        a.wz0 = std::min(a.Dout - 1, need.z0 / scale);
        a.Dout = std::max(a.wz0 + 1, std::min(a.Dout, (need.z1 + scale - 1) / scale)) - a.wz0;
    }
}


// the weights-resident kernel (conv_rw.h) for a 3x3 32 -> 32 layer: window and tile grid as launch_split, one persistent
// workgroup per CU
static int launch_rw(tpz_ctx* ctx, SplitArgs& a, int dil, int epi, double flops) {
    static char names[3][3][96];
    const int di = dil == 1 ? 0 : dil == 2 ? 1 : 2;
    if (!names[di][epi][0])
        snprintf(names[di][epi], sizeof names[di][epi], "conv_split_rw_kernel<K=3x3,D=%d,MT=32,EPI=%d> (weights resident)", dil, epi);
    if (a.wy1 < 0) {
        a.wy1 = -a.wy1;
        flops *= (double)(a.wy1 - a.wy0) * (a.wx1 - a.wx0) / ((double)a.Hout * a.Wout);
    } else {
        a.wy0 = a.wx0 = 0; a.wy1 = a.Hout; a.wx1 = a.Wout;
    }
    a.tiles_x = (a.wx1 - a.wx0 + 31) / 32;
    a.tiles_y = (a.wy1 - a.wy0 + 8 * dil - 1) / (8 * dil) * dil;
    const long long nt = (long long)a.tiles_x * a.tiles_y;
    if (nt >= (1LL << 30)) return fail(ctx, "conv grid too large");
    a.n_tiles = (int)nt;
    if ((size_t)a.cells_in * a.Hin * a.Win * 16 >= ((size_t)1 << 32) - 16)
        return fail(ctx, "image too large for one launch (%d x %d): process it in patches", a.Hin, a.Win);
    const int wgs = std::max(8, ctx->n_cus / 8 * 8);
    const double wy = a.wy1 - a.wy0, wx = a.wx1 - a.wx0, span = 2.0 * dil;
    double bytes = (double)a.cells_in * 32.0 * std::min((double)a.Hin, wy + span) * std::min((double)a.Win, wx + span) +
                   32.0 * a.cells_out * wy * wx * (a.res ? 2.0 : 1.0) + 36864.0;
    const SplitArgs ac = a;
    hipError_t e = enqueue(ctx, 0, flops, names[di][epi], bytes, [=](hipStream_t st) { return launch_conv_rw(ac, dil, epi, wgs, st); });
    if (e != hipSuccess) return fail(ctx, "conv_rw launch failed: %s", hipGetErrorString(e));
    return 0;
}

// one conv layer on the 2xf16 path: split source (and residual), split output or fused fp32 head
// (fold: the input of a folded 1x1 projection, split cells -- the layer then runs ks_fold with the projection's channels
// appended to its K loop, no residual, eval-BN already inside weights and bias)
static int run_conv_split(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, const Slot* sres, Slot& dst,
                          const Slot* s2 = nullptr, bool pooled = false, const Slot* fold = nullptr) {
    const tpz_layer& L = rt.L;
    const SplitKernelInfo& ks = fold ? *rt.ks_fold : pooled ? *rt.ks_pool : *rt.ks;
    SplitArgs a;
    memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const uint4*>(s1.p);
    a.wpk = reinterpret_cast<const uint4*>(fold ? rt.d_wfold : rt.d_wsplit);
    a.wscale = fold ? rt.d_wscale_fold : rt.d_wscale;
    a.bias = bias_view(ctx, fold ? rt.d_bias_fold : rt.d_bias);
    a.res = sres ? reinterpret_cast<const uint4*>(sres->p) : nullptr;
    a.post_scale = fold ? nullptr : rt.d_post_scale;
    a.post_shift = fold ? nullptr : bias_view(ctx, rt.d_post_shift);
    a.head_w = rt.d_head_w;
    a.head_b = ctx->scaled_pass ? 0.f : rt.head_b;
    if (L.head) a.head_out = dst.p;
    else if (ks.epi == EPI_PLAIN_F32) a.out_f32 = dst.p;
    else a.out = reinterpret_cast<uint4*>(dst.p);
    a.zeros = ctx->d_zeros;
    a.flag = ctx->d_flag;
    a.slope = L.slope;
    a.cells_in1 = (int)split_cells(s1.C);
    a.H1 = s1.H; a.W1 = s1.W;
    if (s2) {
        a.in2 = reinterpret_cast<const uint4*>(s2->p);
        a.cells_in = a.cells_in1 + (int)split_cells(s2->C);
        a.Hin = s2->H; a.Win = s2->W;
    } else {
        a.cells_in = a.cells_in1;
        a.Hin = s1.H; a.Win = s1.W;
    }
    a.Cout = L.cout;
    a.cells_out = (int)split_cells(L.cout);
    a.Hout = dst.H; a.Wout = dst.W;
    if (pooled) {          // dst is the pooled tensor; the launch covers the un-pooled conv output
        const Slot& g = s2 ? *s2 : s1;
        a.Hout = g.H + 2 * L.pad - L.dil * (L.k - 1);
        a.Wout = g.W + 2 * L.pad - L.dil * (L.k - 1);
    }
    a.pad_x = a.pad_y = L.pad;
    a.os = 1;
    a.Hfull = dst.H; a.Wfull = dst.W;
    if (sres) { a.Hres = sres->H; a.Wres = sres->W; a.res_crop = L.res_crop; }
    a.n_chunks = rt.s_n_chunks;
    a.cog_inner = L.head ? rt.s_n_cog : 1;
    if (L.dims == 3) {
        a.KZ = L.k; a.pad_z = L.pad; a.Din = s1.D; a.Dout = dst.D; a.Dfull = dst.D; a.Dres = sres ? sres->D : 1; a.ooz = 0;
    }
    double flops = 2.0 * L.cout * L.cin * std::pow((double)L.k, L.dims) * (double)dst.D * a.Hout * a.Wout;
    if (fold) {
        // out(y, x) += proj(h)(y + res_crop, x + res_crop); the centre tap of output y sits at tile-input row y - pad + (k/2) dil
        a.in2 = reinterpret_cast<const uint4*>(fold->p);
        a.fold_cells = rt.fold_cells;
        a.cells_in = a.cells_in1 + rt.fold_cells;
        a.fold_tap = (L.k * L.k) / 2;
        a.in2_H = fold->H; a.in2_W = fold->W;
        a.in2_oy = a.in2_ox = L.res_crop + L.pad - (L.k / 2) * L.dil;
        a.n_chunks = rt.f_n_chunks;
        flops += 2.0 * L.cout * (8.0 * rt.fold_cells) * (double)a.Hout * a.Wout;
        set_window(a, dst.need);
        return launch_split(ctx, ks, a, rt.f_n_cog, flops);
    }
    set_window(a, dst.need);           // (a pooled dst keeps its need in the coordinates of the un-pooled conv output)
    if (rt.d_w_rw && !pooled && !s2 && !ctx->rec_on && ks.epi <= EPI_RES_POST && L.dims == 2 && ctx->rw_enabled) {
        a.wpk = reinterpret_cast<const uint4*>(rt.d_w_rw);
        a.wscale = rt.d_ws_rw;
        return launch_rw(ctx, a, L.dil, ks.epi, flops);
    }
    return launch_split(ctx, ks, a, rt.s_n_cog, flops);
}

// the K-loop schedule of this launch (SplitArgs::plan): tile-invariant, so one table per (kernel, cells, sources) serves every
// launch of the layer; the first launch builds and uploads it (a blocking copy, once)
static const SplitStep* split_plan(tpz_ctx* ctx, const SplitKernelInfo& ks, const SplitArgs& a, bool* next_ok) {
    SplitPlanKey k;
    memset(&k, 0, sizeof k);
    k.cells_in = a.cells_in; k.cells_in1 = a.cells_in1; k.n_chunks = a.n_chunks; k.has_in2 = a.in2 != nullptr;
    k.vol = (a.KZ > 1 || a.Din > 1) ? 1 : 0; k.KZ = a.KZ; k.fold_cells = a.fold_cells; k.fold_tap = a.fold_tap;
    k.srcmajor = (k.vol && a.in2) ? a.vol_srcmajor : 0;
    for (auto& e : ctx->split_plans)
        if (e.ks == &ks && memcmp(&e.key, &k, sizeof k) == 0) { *next_ok = e.next_ok; return e.d; }
    std::vector<SplitStep> h;
    ks.make_plan(k, h);
    *next_ok = (h[0].dma & SPLIT_DMA_NEXT) != 0;
    SplitStep* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(SplitStep)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(SplitStep), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    ctx->split_plans.push_back({&ks, k, d, *next_ok});
    return d;
}

static int launch_split(tpz_ctx* ctx, const SplitKernelInfo& ks, SplitArgs& a, int n_cog, double flops) {
    if (a.wy1 < 0) {
        a.wy1 = -a.wy1;
        flops *= (double)(a.wy1 - a.wy0) * (a.wx1 - a.wx0) / ((double)a.Hout * a.Wout);
        if (a.Dlat > 0) flops *= (double)a.Dout / a.Dlat;
    } else {
        a.wy0 = a.wx0 = 0; a.wy1 = a.Hout; a.wx1 = a.Wout;
    }
    a.tiles_x = (a.wx1 - a.wx0 + ks.TW - 1) / ks.TW;
    a.tiles_y = (a.wy1 - a.wy0 + ks.TH * ks.D - 1) / (ks.TH * ks.D) * ks.D;
    a.xcd_swizzle = 1;
    a.issuer_half = ks.WAVES == 8 && ks.MT >= 96 && !g_no_issuer;   // -3 .. -4 % on the 128-channel tiles, nothing at 64 (tools/split_ablate.hip)
    if (a.KZ < 1) { a.KZ = 1; a.pad_z = 0; a.Din = a.Dout = a.Dfull = a.Dres = 1; a.ooz = 0; }     // 2-D launch
    if (a.Dres < 1) a.Dres = 1;
    a.ncz = n_cog / a.cog_inner;
    const long long gz = (long long)a.ncz * a.Dout * std::max(a.nphase, 1);
    if (a.tiles_y > 65535 || gz > 65535) return fail(ctx, "conv grid too large");
    // the LDS-DMA addresses are 32-bit byte offsets from a wave-uniform base: a chunk of cells (2-D) or one half of the
    // whole tensor (plane-stacked 3-D) must stay below 4 GiB
    if (a.Din > 1 && (size_t)a.cells_in * a.Din * a.Hin * a.Win * 16 >= ((size_t)1 << 32) - 16)
        return fail(ctx, "3-D tensor too large for the plane-stacked 2xf16 kernel (tile the volume)");
    if ((size_t)ks.CC * std::max((size_t)a.Hin * a.Win, (size_t)a.H1 * a.W1) * 16 >= ((size_t)1 << 32) - 16)
        return fail(ctx, "image too large for one launch (%d x %d): process it in patches", a.Hin, a.Win);
    // ... and so are the epilogue's buffer offsets: the two cell planes of a channel fragment, of the output and of the residual
    if (ks.epi != EPI_HEAD && ks.epi != EPI_PLAIN_F32 &&
        2 * std::max((size_t)a.Dfull * a.Hfull * a.Wfull, (size_t)a.Dres * a.Hres * a.Wres) * 16 >= ((size_t)1 << 32) - 16)
        return fail(ctx, "tensor too large for one launch (%d x %d x %d): process it in patches", a.Dfull, a.Hfull, a.Wfull);
    dim3 grid(a.tiles_x, a.tiles_y, (unsigned)gz);
    bool next_ok = false;
    a.plan = split_plan(ctx, ks, a, &next_ok);
    if (!a.plan) return fail(ctx, "out of device memory (K-loop plan)");
    // Persistent workgroups (conv_split.h MODE 4): a few per CU, each walking its share of the tiles and prefetching its next
    // tile's first chunk during the current tile's last -- for plain single-source layers with several tiles per workgroup.
    // Not under the patch lanes: a persistent grid holds every CU until it ends, and the lanes live on the small launches of one
    // patch slipping in beside the large ones of its neighbour.
    a.n_tiles = 0;
    {
        const long long nt = (long long)a.tiles_x * a.tiles_y * gz;
        const int slots = ctx->n_cus * (ks.WAVES == 8 ? 1 : 2);
        const bool plain = !a.in2 && a.KZ <= 1 && a.Din <= 1;
        const bool eligible = next_ok && plain && ks.epi != EPI_HEAD && a.cog_inner == 1 && nt < (1LL << 30);
        // (measured, profiles/r03_persistent_ab.txt: +5 .. +60 % on the tiles of up to 96 channels, whose prologue is 10 - 20 % of
        // a tile; +-0 on the 128-channel 8-wave tiles, where the longer scalar state costs the K loop what the prologue gave)
        const bool want = ctx->persist_mode == 2 || (ctx->persist_mode == 1 && !ctx->lanes_on && nt >= 2LL * slots && ks.MT <= 96);
        if (eligible && want && !ctx->rec_on) {
            const int wgs = ctx->persist_wgs > 0 ? ctx->persist_wgs : slots;
            a.n_tiles = (int)nt;
            grid = dim3((unsigned)std::max(8, wgs / 8 * 8), 1, 1);
        }
    }
    // algorithmic HBM bytes of the launch: the input window (with its halo) of every source once, the weights once, the output
    // window once (+ the residual it adds); 4 bytes per element in either format
    double bytes = 0;
    {
        const double wy = a.wy1 - a.wy0, wx = a.wx1 - a.wx0, span = (double)ks.D * (ks.K - 1), spanx = (double)ks.D * (ks.KX - 1);
        const double planes = (double)a.Dout * std::max(a.nphase, 1);
        bytes += (double)a.cells_in * 32.0 * std::min((double)a.Hin, wy + span) * std::min((double)a.Win, wx + spanx) * (a.Din > 1 ? a.Din : 1);
        const double outpx = wy * wx * planes * (a.os > 1 && a.nphase == 0 && a.subpix_cout > 0 ? 4.0 : 1.0);
        bytes += (a.out_f32 ? 4.0 * a.Cout : a.head_out ? 4.0 : 32.0 * a.cells_out) * outpx;
        if (a.res) bytes += 32.0 * a.cells_out * outpx;
        bytes += (double)n_cog * ks.stages(a.cells_in * std::max(a.KZ, 1)) * ks.W_STEP_BYTES * std::max(a.nphase, 1);
    }
    // patch raster (conv_split.h, xcd_swizzle 2) for the one-workgroup-per-CU tiles of a launch of its own: the grid is padded to
    // whole 8 x 4 blocks of tiles
    if (ctx->raster && !ctx->rec_on && a.n_tiles == 0 && ks.WAVES == 8 && (long long)a.tiles_x * a.tiles_y >= 512) {
        a.xcd_swizzle = 2;
        grid = dim3((unsigned)((a.tiles_x + 7) / 8 * 8), (unsigned)((a.tiles_y + 3) / 4 * 4), (unsigned)gz);
    }
    if (ctx->rec_on) {
        // a batched pass: recorded; rec_flush issues it together with the same layer's launch of the other images
        RecOp op;
        op.ks = &ks;
        op.a = a;
        op.a.n_tiles = (int)std::min<long long>((long long)a.tiles_x * a.tiles_y * gz, 0x7fffffff);
        op.grid = grid;
        op.cls = 0; op.flops = flops; op.bytes = bytes; op.key = ks.name;
        ctx->rec[ctx->rec_cur].push_back(std::move(op));
        return 0;
    }
    prof_begin(ctx, 0, flops, ks.name, bytes);
    hipError_t e = ks.launch(a, grid, ctx->stream);
    prof_end(ctx);
    ++ctx->n_launches;
    HIPCHK(ctx, e);
    return 0;
}

// conv(cat(upsample2x(s1), s2)) on the 2xf16 path (prepare_split_phases).  s1: split; s2: fp32 when it is the
// 1-channel image (stem kernel), else split; dst: split.
static int run_conv_split_phases(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, const Slot& s2, Slot& dst) {
    const tpz_layer& L = rt.L;
    const LayerRT::SplitPhase& sp = rt.sphase;
    const LayerRT::Phase& ph = rt.phase;
    if (sp.sub_with_skip) {
        // one plain sub-pixel launch: low-resolution source + the space-to-depth copy of the 1-channel skip source
        if (s2.pitch != s2.W || s2.ps != (long long)s2.H * s2.W) return fail(ctx, "2xf16 decoder needs a dense skip source");
        float* X = (float*)pool_alloc(ctx, (size_t)8 * s1.H * s1.W * sizeof(float));
        if (!X) return fail(ctx, "out of device memory");
        hipError_t e;
        {
            const float* sp_ = s2.p; unsigned* fl_ = ctx->d_flag;
            const int h1 = s1.H, w1 = s1.W, h2 = s2.H, w2 = s2.W;
            e = enqueue(ctx, [=](hipStream_t st) { return launch_s2d_split(sp_, X, 1, h1, w1, h2, w2, 2, fl_, st); });
        }
        if (e != hipSuccess) { pool_release(ctx, X); return fail(ctx, "s2d failed: %s", hipGetErrorString(e)); }
        SplitArgs a;
        memset(&a, 0, sizeof a);
        a.in = reinterpret_cast<const uint4*>(s1.p);
        a.in2 = reinterpret_cast<const uint4*>(X);
        a.wpk = reinterpret_cast<const uint4*>(sp.d_w_low);
        a.wscale = sp.d_ws_low;
        a.bias = bias_view(ctx, rt.d_bias);
        a.subpix_cout = L.cout;
        a.pad_x = a.pad_y = 1;
        a.out = reinterpret_cast<uint4*>(dst.p);
        a.zeros = ctx->d_zeros;
        a.flag = ctx->d_flag;
        a.slope = L.slope;
        a.cells_in1 = (int)split_cells(s1.C);
        a.cells_in = a.cells_in1 + 1;
        a.Hin = a.H1 = s1.H; a.Win = a.W1 = s1.W;
        a.Cout = L.cout; a.cells_out = (int)split_cells(L.cout);
        a.Hout = s1.H; a.Wout = s1.W;
        a.os = 2;
        a.Hfull = dst.H; a.Wfull = dst.W;
        a.n_chunks = sp.n_chunks_low;
        a.cog_inner = 1;
        const double fl = 2.0 * L.cout * (ph.c1 * 9.0 * 4.0 + 25.0 * 4.0) * (double)s1.H * s1.W;
        set_window(a, dst.need, 2);
        const int rc = launch_split(ctx, *sp.ks_sub, a, sp.n_cog_sub, fl);
        pool_release(ctx, X);
        return rc;
    }
    // ---- skip-source part over the full grid: bias, no activation
    float* Xs2d = nullptr;
    if (sp.low_with_skip) {
        if (s2.pitch != s2.W || s2.ps != (long long)s2.H * s2.W) return fail(ctx, "2xf16 decoder needs a dense skip source");
        Xs2d = (float*)pool_alloc(ctx, (size_t)8 * s1.D * s1.H * s1.W * sizeof(float));
        if (!Xs2d) return fail(ctx, "out of device memory");
        hipError_t e;
        {
            const float* sp_ = s2.p; unsigned* fl_ = ctx->d_flag;
            const int d1 = s1.D, h1 = s1.H, w1 = s1.W, h2 = s2.H, w2 = s2.W, dims = L.dims;
            e = enqueue(ctx, [=](hipStream_t st) { return launch_s2d_split(sp_, Xs2d, d1, h1, w1, h2, w2, dims, fl_, st); });
        }
        if (e != hipSuccess) { pool_release(ctx, Xs2d); return fail(ctx, "s2d failed: %s", hipGetErrorString(e)); }
    } else if (sp.ki_skip_stem) {
        ConvArgs a;
        memset(&a, 0, sizeof a);
        a.in = s2.p;
        a.wpk = ph.d_w_skip;
        a.bias = bias_view(ctx, rt.d_bias);
        a.out = dst.p;
        a.zeros = ctx->d_zeros;
        a.flag = ctx->d_flag;
        a.Cin = a.Cin1 = 1;
        a.Din = a.D1 = s2.D; a.Hin = a.H1 = s2.H; a.Win = a.W1 = s2.W;
        a.cs1 = s2.cs; a.ps1 = s2.ps; a.pitch1 = s2.pitch;
        a.Cout = L.cout;
        a.Dout = dst.D; a.Hout = dst.H; a.Wout = dst.W;
        a.pad = a.pad_x = a.pad_y = a.pad_z = L.pad;
        a.os = 1;
        a.Dfull = dst.D; a.Hfull = dst.H; a.Wfull = dst.W;
        a.slope = 1.f;
        a.n_chunks = 1;
        a.cog_inner = 1;
        const double fl = 2.0 * L.cout * std::pow((double)L.k, L.dims) * (double)dst.D * dst.H * dst.W;
        if (launch_mfma(ctx, *sp.ki_skip_stem, a, 1, fl)) return 1;
    } else {
        SplitArgs a;
        memset(&a, 0, sizeof a);
        a.in = reinterpret_cast<const uint4*>(s2.p);
        a.wpk = reinterpret_cast<const uint4*>(sp.d_w_skip);
        a.wscale = sp.d_ws_skip;
        a.bias = bias_view(ctx, rt.d_bias);
        a.out = reinterpret_cast<uint4*>(dst.p);
        a.zeros = ctx->d_zeros;
        a.flag = ctx->d_flag;
        a.slope = 1.f;
        a.cells_in = a.cells_in1 = (int)split_cells(s2.C);
        a.Hin = a.H1 = s2.H; a.Win = a.W1 = s2.W;
        a.Cout = L.cout; a.cells_out = (int)split_cells(L.cout);
        a.Hout = dst.H; a.Wout = dst.W;
        a.pad_x = a.pad_y = L.pad;
        a.os = 1; a.Hfull = dst.H; a.Wfull = dst.W;
        if (L.dims == 3) { a.KZ = L.k; a.pad_z = L.pad; a.Din = s2.D; a.Dout = a.Dfull = dst.D; a.Dres = 1; }
        a.n_chunks = sp.n_chunks_skip;
        a.cog_inner = 1;
        const double fl = 2.0 * L.cout * ph.c2 * std::pow((double)L.k, L.dims) * (double)dst.D * dst.H * dst.W;
        set_window(a, dst.need);       // (even-aligned by need_regions: the parity launch below adds itself in place)
        if (launch_split(ctx, *sp.ks_skip, a, sp.n_cog_skip, fl)) return 1;
    }
    // ---- every output parity over the low-resolution source in one launch, added in place, then the activation
    {
        SplitArgs a;
        memset(&a, 0, sizeof a);
        a.in = reinterpret_cast<const uint4*>(s1.p);
        a.wpk = reinterpret_cast<const uint4*>(sp.d_w_low);
        a.wscale = sp.d_ws_low;
        if (sp.ks_sub) {
            a.subpix_cout = L.cout;
            a.pad_x = a.pad_y = 1;
        } else {
            a.nphase = 1 << L.dims;
            a.phase_k = L.k;
            a.w_phase_bytes = sp.w_phase_bytes;
            a.ws_phase_stride = (int)chan_pad(L.cout);
        }
        a.out = reinterpret_cast<uint4*>(dst.p);
        a.res = reinterpret_cast<const uint4*>(dst.p);
        a.zeros = ctx->d_zeros;
        a.flag = ctx->d_flag;
        a.slope = L.slope;
        a.cells_in = a.cells_in1 = (int)split_cells(s1.C);
        if (sp.low_with_skip) {                        // + the space-to-depth cell of the skip source; plain epilogue
            a.in2 = reinterpret_cast<const uint4*>(Xs2d);
            a.cells_in = a.cells_in1 + 1;
            a.res = nullptr;
            a.bias = bias_view(ctx, rt.d_bias);
            a.vol_srcmajor = sp.srcmajor ? 1 : 0;
        }
        a.Hin = a.H1 = s1.H; a.Win = a.W1 = s1.W;
        a.Cout = L.cout; a.cells_out = (int)split_cells(L.cout);
        a.Hout = s1.H; a.Wout = s1.W;                  // the lattice of one parity
        a.os = 2;
        a.Hfull = dst.H; a.Wfull = dst.W;
        a.Hres = dst.H; a.Wres = dst.W; a.res_crop = 0;
        a.KZ = 1; a.Din = a.Dout = a.Dfull = a.Dres = 1;
        if (L.dims == 3) { a.KZ = ph.k1; a.Din = s1.D; a.Dout = s1.D; a.Dfull = dst.D; a.Dres = dst.D; }
        a.n_chunks = sp.n_chunks_low;
        a.cog_inner = 1;
        const double fl = 2.0 * L.cout * ph.c1 * std::pow((double)ph.k1, L.dims) * (double)s1.D * s1.H * s1.W * (1 << L.dims);
        const SplitKernelInfo& kk = sp.ks_sub ? *sp.ks_sub : (sp.low_with_skip ? *sp.ks_low_plain : *sp.ks_low);
        set_window(a, dst.need, 2);
        const int rc = launch_split(ctx, kk, a, sp.ks_sub ? sp.n_cog_sub : sp.n_cog_low, fl);
        if (Xs2d) pool_release(ctx, Xs2d);
        if (rc) return 1;
    }
    return 0;
}

// 1-channel stem on the 2xf16 path: x-shifted copy of the image (kx taps as channels), then a k x 1 column kernel
static int run_stem_split(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, Slot& dst, bool pooled = false) {
    const tpz_layer& L = rt.L;
    const SplitKernelInfo& ks = pooled ? *rt.ks_pool : *rt.ks_stem;
    if (s1.pitch != s1.W || s1.ps != (long long)s1.H * s1.W) return fail(ctx, "2xf16 stem needs a dense input");
    const int ncell = (L.k + 7) / 8;
    const size_t rows = (size_t)s1.D * s1.H;
    // conv output geometry (dst is the pooled tensor when the max-pool is fused)
    const int Hc = s1.H + 2 * L.pad - (L.k - 1), Wc = s1.W + 2 * L.pad - (L.k - 1);
    float* X = (float*)pool_alloc(ctx, (size_t)ncell * 8 * rows * Wc * sizeof(float));
    if (!X) return fail(ctx, "out of device memory");
    // (2-D with a window: only the rows and columns the windowed conv reads -- output row y reads input rows y - pad .. y + pad)
    const Rect& w = dst.need;
    hipError_t e;
    {
        const float* sp_ = s1.p; unsigned* fl_ = ctx->d_flag;
        const int k = L.k, pad = L.pad, W1 = s1.W;
        if (w.on && L.dims == 2) {
            const size_t r0 = (size_t)std::max(0, w.y0 - L.pad), r1 = (size_t)std::min(s1.H, w.y1 + L.pad);
            const int c0 = w.x0, c1 = std::min(Wc, (w.x1 + 1) & ~1);
            e = enqueue(ctx, [=](hipStream_t st) { return launch_shiftx_split(sp_, X, k, pad, rows, W1, Wc, fl_, st, r0, r1, c0, c1); });
        } else {
            e = enqueue(ctx, [=](hipStream_t st) { return launch_shiftx_split(sp_, X, k, pad, rows, W1, Wc, fl_, st); });
        }
    }
    if (e != hipSuccess) { pool_release(ctx, X); return fail(ctx, "shiftx failed: %s", hipGetErrorString(e)); }
    SplitArgs a;
    memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const uint4*>(X);
    a.wpk = reinterpret_cast<const uint4*>(rt.d_wsplit);
    a.wscale = rt.d_wscale;
    a.bias = bias_view(ctx, rt.d_bias);
    a.out = reinterpret_cast<uint4*>(dst.p);
    a.zeros = ctx->d_zeros;
    a.flag = ctx->d_flag;
    a.slope = L.slope;
    a.cells_in = a.cells_in1 = ncell;
    a.Hin = a.H1 = s1.H; a.Win = a.W1 = Wc;
    a.Cout = L.cout; a.cells_out = (int)split_cells(L.cout);
    a.Hout = Hc; a.Wout = Wc;
    a.pad_x = 0; a.pad_y = L.pad;
    a.os = 1; a.Hfull = dst.H; a.Wfull = dst.W;
    if (L.dims == 3) { a.KZ = L.k; a.pad_z = L.pad; a.Din = s1.D; a.Dout = a.Dfull = dst.D; a.Dres = 1; }
    a.n_chunks = rt.s_n_chunks;
    a.cog_inner = 1;
    const double fl = 2.0 * L.cout * std::pow((double)L.k, L.dims) * (double)dst.D * Hc * Wc;
    set_window(a, dst.need);
    const int rc = launch_split(ctx, ks, a, rt.s_n_cog, fl);
    pool_release(ctx, X);
    return rc;
}

// 1-output-channel last conv on the 2xf16 path: k virtual output channels (one per kx tap) over W + 2*pad columns
// by a k x 1 column kernel storing fp32, then out[x] = sum_v Y[v][x + v] + bias (and the un-normalisation)
static int run_last_split(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, Slot& dst, const float* d_nrm, int norm_out,
                          const Slot* sres = nullptr) {
    const tpz_layer& L = rt.L;
    if (rt.d_wlast) {
        // one pass: stencil + bias + residual + un-normalisation (conv_cout1_split_kernel)
        const Rect& w = dst.need;
        const int z0 = w.on ? w.z0 : 0, z1 = w.on ? std::min(dst.D, w.z1) : dst.D;
        const int y0 = w.on ? w.y0 : 0, y1 = w.on ? std::min(dst.H, w.y1) : dst.H;
        const int x0 = w.on ? w.x0 : 0, x1 = w.on ? std::min(dst.W, w.x1) : dst.W;
        const double vox = (double)(z1 - z0) * (y1 - y0) * (x1 - x0);
        const double taps = std::pow((double)L.k, L.dims);
        const double fl = 2.0 * L.cin * taps * vox;
        // algorithmic bytes: the input box (with its halo) once, the output (and the residual) once, the weights once
        const double by = 32.0 * split_cells(s1.C) * (double)std::min(dst.D, z1 - z0 + (L.dims == 3 ? 2 * L.pad : 0)) *
                              std::min(dst.H, y1 - y0 + 2 * L.pad) * std::min(dst.W, x1 - x0 + 2 * L.pad) +
                          4.0 * vox * (sres ? 2 : 1) + 4.0 * L.cin * taps;
        const void* ip = s1.p; const float* wp_ = rt.d_wlast; float* op = dst.p;
        const float* resp = sres ? sres->p : nullptr;
        const float b0 = L.b_off >= 0 ? rt.bias0 : 0.f;
        const int cells = (int)split_cells(s1.C), k = L.k, kz = L.dims == 3 ? L.k : 1, Dd = dst.D, Hd = dst.H, Wd = dst.W;
        hipError_t e = enqueue(ctx, 0, fl, "conv_cout1_split_kernel (last conv: fp32 stencil on the vector ALUs + bias + un-normalisation)",
                               by, [=](hipStream_t st) {
                                   return launch_conv_cout1_split(ip, wp_, op, resp, d_nrm, norm_out, b0, cells, k, kz, Dd, Hd, Wd, z0, z1,
                                                                  y0, y1, x0, x1, st);
                               });
        if (e != hipSuccess) return fail(ctx, "conv_cout1_split failed: %s", hipGetErrorString(e));
        return 0;
    }
    const SplitKernelInfo& ks = *rt.ks_last;
    const int Wp = dst.W + 2 * L.pad;
    const size_t rows = (size_t)dst.D * dst.H;
    float* Y = (float*)pool_alloc(ctx, (size_t)L.k * rows * Wp * sizeof(float));
    if (!Y) return fail(ctx, "out of device memory");
    SplitArgs a;
    memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const uint4*>(s1.p);
    a.wpk = reinterpret_cast<const uint4*>(rt.d_wsplit);
    a.wscale = rt.d_wscale;
    a.out_f32 = Y;
    a.zeros = ctx->d_zeros;
    a.flag = ctx->d_flag;
    a.slope = 1.f;
    a.cells_in = a.cells_in1 = (int)split_cells(s1.C);
    a.Hin = a.H1 = s1.H; a.Win = a.W1 = s1.W;
    a.Cout = L.k; a.cells_out = 1;
    a.Hout = dst.H; a.Wout = Wp;
    a.pad_x = a.pad_y = L.pad;
    a.os = 1; a.Hfull = dst.H; a.Wfull = Wp;
    if (L.dims == 3) { a.KZ = L.k; a.pad_z = L.pad; a.Din = s1.D; a.Dout = a.Dfull = dst.D; a.Dres = 1; }
    a.n_chunks = rt.s_n_chunks;
    a.cog_inner = 1;
    const double fl = 2.0 * L.cin * std::pow((double)L.k, L.dims) * (double)dst.D * dst.H * dst.W;
    const Rect& w = dst.need;
    set_window(a, w, 1, 2 * L.pad);       // Y columns x .. x + k - 1 feed output column x
    int rc = launch_split(ctx, ks, a, rt.s_n_cog, fl);
    if (!rc) {
        // (labelled: an HBM-bound kernel whose bandwidth bench.py reports -- reads k planes of Wp columns, writes one of W)
        const double ss_rows = w.on ? (double)(w.y1 - w.y0) * (L.dims == 3 ? w.z1 - w.z0 : 1) : (double)rows, ss_cols = w.on ? (double)(w.x1 - w.x0) : (double)dst.W;
        // (a residual of the output's own size -- UDenoiseNet3: x - dec1(h), weights negated -- is added here, in fp32)
        const float* resp = sres ? sres->p : nullptr;
        float* dp_ = dst.p;
        const int k = L.k, Wd = dst.W;
        const float b0 = L.b_off >= 0 ? rt.bias0 : 0.f;
        const size_t r0 = w.on ? (size_t)w.y0 : 0, r1 = w.on ? (size_t)w.y1 : (size_t)-1;
        const int c0 = w.on ? w.x0 : 0, c1 = w.on ? w.x1 : 0x7fffffff;
        const int Hp = (w.on && L.dims == 3) ? dst.H : 0, z0 = w.z0, z1 = std::min(dst.D, w.z1);
        hipError_t e = enqueue(ctx, 2, 0.0, "shiftsum (last conv: sum of the k column-kernel planes + bias + un-normalisation)",
                               4.0 * ss_rows * ((double)L.k * (ss_cols + 2 * L.pad) + ss_cols), [=](hipStream_t st) {
                                   return launch_shiftsum(Y, dp_, k, rows, Wd, Wp, b0, d_nrm, norm_out, st, r0, r1, c0, c1, resp, Hp, z0, z1);
                               });
        if (e != hipSuccess) rc = fail(ctx, "shiftsum failed: %s", hipGetErrorString(e));
    }
    pool_release(ctx, Y);
    return rc;
}

// the slot's 2-D tensor in the wanted format: the producer's own buffer, or a converted copy made once
static float* slot_as(tpz_ctx* ctx, Slot& s, bool want_split) {
    if (s.split == want_split) return s.p;
    if (s.alt) return s.alt;
    if (s.pitch != s.W || s.ps != (long long)s.H * s.W || s.cs != s.ps * s.D) return nullptr;
    const size_t c_alloc = want_split ? split_cells(s.C) * 8 : (size_t)s.C;
    float* q = (float*)pool_alloc(ctx, c_alloc * s.D * s.H * s.W * sizeof(float));
    if (!q) return nullptr;
    // cells are [c/8][D*H*W]: a volume converts as an image of D*H rows
    hipError_t e;
    {
        const float* sp_ = s.p; unsigned* fl_ = ctx->d_flag;
        const int C = s.C, R = s.D * s.H, W = s.W;
        e = enqueue(ctx, [=](hipStream_t st) {
            return want_split ? launch_to_split(sp_, q, C, R, W, fl_, st) : launch_from_split(sp_, q, C, R, W, st);
        });
    }
    if (e != hipSuccess) { pool_release(ctx, q); return nullptr; }
    s.alt = q;
    return q;
}

static int run_conv(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, const Slot* s2, const Slot* sres, Slot& dst,
                    const float* d_nrm, int norm_out, bool split_out = false) {
    const tpz_layer& L = rt.L;
    ConvArgs a;
    memset(&a, 0, sizeof a);
    a.in = s1.p;
    a.in2 = s2 ? s2->p : nullptr;
    a.wpk = rt.d_wpk;
    a.bias = bias_view(ctx, rt.d_bias);
    a.res = sres ? sres->p : nullptr;
    a.post_scale = rt.d_post_scale;
    a.post_shift = bias_view(ctx, rt.d_post_shift);
    a.head_w = rt.d_head_w;
    a.head_b = ctx->scaled_pass ? 0.f : rt.head_b;
    a.nrm = d_nrm;
    a.zeros = ctx->d_zeros;
    a.norm_out = d_nrm ? norm_out : 0;
    a.Cin = L.cin;
    a.Cin1 = s1.C;
    const Slot& geo = s2 ? *s2 : s1;
    a.Din = geo.D; a.Hin = geo.H; a.Win = geo.W;
    a.D1 = s1.D; a.H1 = s1.H; a.W1 = s1.W;
    a.cs1 = s1.cs; a.ps1 = s1.ps; a.pitch1 = s1.pitch;
    if (s2) { a.cs2 = s2->cs; a.ps2 = s2->ps; a.pitch2 = s2->pitch; }
    a.Cout = L.cout;
    a.Dout = dst.D; a.Hout = dst.H; a.Wout = dst.W;
    a.pad = L.pad;
    a.pad_x = a.pad_y = a.pad_z = L.pad;
    a.os = 1;
    a.Dfull = dst.D; a.Hfull = dst.H; a.Wfull = dst.W;
    if (sres) { a.Dres = sres->D; a.Hres = sres->H; a.Wres = sres->W; a.res_crop = L.res_crop; }
    a.slope = L.slope;
    if (L.head) { a.head_out = dst.p; a.out = nullptr; }
    else a.out = dst.p;
    set_window(a, dst.need);          // patched / tiled denoise: only what the kept centre depends on (need_regions)
    const double flops = 2.0 * L.cout * L.cin * std::pow((double)L.k, L.dims) * (double)dst.D * dst.H * dst.W;
    if (rt.ki) {
        const ConvKernelInfo& ki = split_out ? *rt.ki_stem_split : *rt.ki;    // same tile and weight packing
        a.flag = ctx->d_flag;
        const LayerRT::Phase& ph = rt.phase;
        if (ph.valid && s2 && s1.C == ph.c1 && s2->C == ph.c2 && s2->H == 2 * s1.H && s2->W == 2 * s1.W &&
            (L.dims == 2 || s2->D == 2 * s1.D))
            return run_conv_phases(ctx, rt, a, s1, *s2, dst);
        if (s2 && (s1.C % ki.NCH) != 0)
            return fail(ctx, "fused concat needs the first source's channels (%d) to be a multiple of %d", s1.C, ki.NCH);
        a.n_chunks = rt.n_chunks;
        a.cog_inner = rt.cog_inner;
        if (launch_mfma(ctx, ki, a, rt.n_cog, flops)) return 1;
    } else {
        if (s1.D != geo.D || s1.H != geo.H || s1.W != geo.W) return fail(ctx, "direct conv cannot upsample");
        const ConvArgs ac = a;
        const float* wp_ = rt.d_wpk;
        const int k = L.k, kz = L.dims == 3 ? L.k : 1, dil = L.dil;
        hipError_t e = enqueue(ctx, 1, a.wy1 > 0 ? flops * (a.wy1 - a.wy0) * (a.wx1 - a.wx0) / ((double)a.Hout * a.Wout) *
                                                       (a.wz1 > 0 ? (double)(a.wz1 - a.wz0) / a.Dout : 1.0) : flops, nullptr,
                               0.0, [=](hipStream_t st) { return launch_conv_direct(ac, wp_, k, kz, dil, st); });
        HIPCHK(ctx, e);
    }
    return 0;
}

// runs the layer program.  `slots` holds preset external slots (at least slot 0); the dst of the last
// layer is written to d_out (dense).  d_nrm != nullptr: slot 0 is normalised on load wherever it is read
// and the output is un-normalised (Denoise._denoise, topaz/denoise.py:283-295).
// PyTorch 'nearest' source index exactly as the kernels compute it (conv_mfma.h nearest_src)
static int nearest_src_host(int dst, int in_sz, int out_sz) {
    if (in_sz == out_sz) return dst;
    const float scale = (float)in_sz / (float)out_sz;
    const int v = (int)floorf((float)dst * scale);
    return v < in_sz - 1 ? v : in_sz - 1;
}

// Which part of every slot's tensor do the pixels `keep` of the program's output depend on?  (2-D programs, on the 2xf16
// kernels or -- exact mode -- on the fp32 kernels, whose launches take the same windows: ConvArgs::wy0..wx1.)  A patched denoise keeps only the centre of each patch (denoise.py:299-323: patch_size pixels of a patch_size +
// 2*padding tile; CLI default 1024 of 2024), and the U-Net's receptive field (~230 pixels) is far smaller than the default
// padding (500): most of what the full-size layers of a patch compute is thrown away.  Walking the layer list backwards from
// `keep` -- a conv needs its window grown by the padding, a 2x2 max-pool twice the window, a nearest-upsampled source the
// window mapped through the same index formula the kernel uses -- gives every layer the rectangle it has to produce; the
// launches cover just that (SplitArgs::wy0..wx1).  Nothing else changes: the tensors keep their full-size layout and
// coordinates, every kept pixel is computed by the same instructions on the same operands as before (bit-identical output,
// tests/test_gpu_denoise.py), the statistics of the normalisation are still those of the whole padded patch.
// 3-D programs (the tiles of Denoise3D.denoise, denoise.py:340-377: patch_size^3 voxels kept of a (patch_size + 2*padding)^3
// tile -- 1/8 of the tile at the CLI's 96 / 48) are windowed the same way with boxes instead of rectangles, on the 2xf16
// kernels only: the plane-stacked launches take the planes of the box (SplitArgs::wz0, Dout) besides its rectangle.
// Returns an empty vector when the program cannot be windowed (a 3-D program on the fp32 kernels, a 2xf16 program with a layer
// left on an fp32 kernel, an op it does not know).
static std::vector<Rect> need_regions(const tpz_model* m, int D0, int H0, int W0, const Rect& keep, bool split) {
    const int nl = (int)m->layers.size();
    std::vector<Rect> need;
    // (TPZ_TRACE_HOST=1 says which check left a program whole)
    auto bail = [&](int why) {
        if (g_trace_host) fprintf(stderr, "[tpz host] need_regions: program left whole (check %d)\n", why);
        need.clear();
        return need;
    };
    if (!keep.on || !m->ctx->roi_enabled || nl == 0) return need;
    const int dims = D0 > 1 ? 3 : 2;
    // (round 5: the fp32 kernels of a 3-D program take boxes as well -- ConvArgs::wz0 / wz1 -- so exact mode and an overflow
    // re-run of a tiled tomogram no longer compute every tile in full)
    // shapes of all slots
    std::vector<int> Ds(m->n_slots, 1), Hs(m->n_slots, 0), Ws(m->n_slots, 0);
    Ds[0] = D0; Hs[0] = H0; Ws[0] = W0;
    for (int i = 0; i < nl; ++i) {
        const LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        if (L.dims != dims || (split && rt.folded_into >= 0)) return bail(2);
        if (L.op == TPZ_OP_CONV) {
            // (a 2xf16 program with a layer left on an fp32 kernel stays whole: the format conversions between the two read
            // whole tensors, and what a windowed producer did not write may hold any bit pattern -- the overflow flag)
            const int g = L.src2 >= 0 ? L.src2 : L.src, span = L.dil * (L.k - 1);
            // (the per-parity form runs when the skip source is exactly twice the upsampled one -- run_program's rule; with a
            // 1-channel skip source it either takes that source as a space-to-depth cell or runs it through the fp32 stem kernel
            // over the whole grid, which reads only the image)
            const bool parity = rt.sphase.valid && L.src2 >= 0 && Hs[g] == 2 * Hs[L.src] && Ws[g] == 2 * Ws[L.src] &&
                                (dims == 2 || Ds[g] == 2 * Ds[L.src]);
            const bool own = rt.ks || rt.ks_last || (rt.ks_stem && L.src == 0);
            const bool windowed = dims == 3 ? (own || parity) : (own || (rt.sphase.valid && !rt.sphase.ki_skip_stem));
            if (split && !windowed) return bail(3);
            Hs[L.dst] = Hs[g] + 2 * L.pad - span; Ws[L.dst] = Ws[g] + 2 * L.pad - span;
            if (dims == 3) Ds[L.dst] = Ds[g] + 2 * L.pad - span;
        } else if (L.op == TPZ_OP_MAXPOOL2) {
            Hs[L.dst] = Hs[L.src] / 2; Ws[L.dst] = Ws[L.src] / 2;
            if (dims == 3) Ds[L.dst] = Ds[L.src] / 2;
        } else if (L.op == TPZ_OP_MAXPOOL) {
            Hs[L.dst] = Hs[L.src] - L.dil * (L.k - 1); Ws[L.dst] = Ws[L.src] - L.dil * (L.k - 1);
            if (dims == 3) Ds[L.dst] = Ds[L.src] - L.dil * (L.k - 1);
        } else {
            return bail(4);
        }
        if (Ds[L.dst] < 1 || Hs[L.dst] < 1 || Ws[L.dst] < 1) return bail(5);
    }
    need.assign(m->n_slots, Rect());
    // (2-D: every box is the one plane [0, 1))
    auto clip = [&](Rect r, int slot) {
        r.y0 = std::max(0, r.y0); r.x0 = std::max(0, r.x0);
        r.y1 = std::min(Hs[slot], r.y1); r.x1 = std::min(Ws[slot], r.x1);
        if (dims == 3) { r.z0 = std::max(0, r.z0); r.z1 = std::min(Ds[slot], r.z1); }
        else { r.z0 = 0; r.z1 = 1; }
        r.on = true;
        return r;
    };
    need[m->layers[nl - 1].L.dst] = clip(keep, m->layers[nl - 1].L.dst);
    for (int i = nl - 1; i >= 0; --i) {
        const tpz_layer& L = m->layers[i].L;
        Rect R = need[L.dst];
        if (!R.on) { return bail(6); }            // a tensor nobody reads: leave the program alone
        if (L.op == TPZ_OP_CONV) {
            // launch windows start and end on even pixels: the per-parity kernels work on the half-resolution lattice, a
            // fused max-pool pairs rows and columns
            R.y0 &= ~1; R.x0 &= ~1;
            R.y1 = std::min(Hs[L.dst], (R.y1 + 1) & ~1); R.x1 = std::min(Ws[L.dst], (R.x1 + 1) & ~1);
            if (dims == 3) { R.z0 &= ~1; R.z1 = std::min(Ds[L.dst], (R.z1 + 1) & ~1); }
            need[L.dst] = R;
            const int g = L.src2 >= 0 ? L.src2 : L.src, span = L.dil * (L.k - 1);
            Rect G;                                           // in the coordinates of the (upsampled) input grid
            G.y0 = R.y0 - L.pad; G.x0 = R.x0 - L.pad; G.y1 = R.y1 - L.pad + span; G.x1 = R.x1 - L.pad + span;
            if (dims == 3) { G.z0 = R.z0 - L.pad; G.z1 = R.z1 - L.pad + span; }
            G = clip(G, g);
            if (L.src2 >= 0) {
                need[L.src2].unite(G);
                Rect S;                                       // the first source, nearest-upsampled to the grid of the second
                S.y0 = nearest_src_host(G.y0, Hs[L.src], Hs[g]); S.y1 = nearest_src_host(G.y1 - 1, Hs[L.src], Hs[g]) + 1;
                S.x0 = nearest_src_host(G.x0, Ws[L.src], Ws[g]); S.x1 = nearest_src_host(G.x1 - 1, Ws[L.src], Ws[g]) + 1;
                if (dims == 3) { S.z0 = nearest_src_host(G.z0, Ds[L.src], Ds[g]); S.z1 = nearest_src_host(G.z1 - 1, Ds[L.src], Ds[g]) + 1; }
                need[L.src].unite(clip(S, L.src));
            } else {
                need[L.src].unite(G);
            }
            if (L.res >= 0) {
                Rect Q = R;
                Q.y0 += L.res_crop; Q.y1 += L.res_crop; Q.x0 += L.res_crop; Q.x1 += L.res_crop;
                if (dims == 3) { Q.z0 += L.res_crop; Q.z1 += L.res_crop; }
                need[L.res].unite(clip(Q, L.res));
            }
        } else if (L.op == TPZ_OP_MAXPOOL2) {
            Rect Q;
            Q.y0 = 2 * R.y0; Q.x0 = 2 * R.x0; Q.y1 = 2 * R.y1; Q.x1 = 2 * R.x1;
            if (dims == 3) { Q.z0 = 2 * R.z0; Q.z1 = 2 * R.z1; }
            need[L.src].unite(clip(Q, L.src));
        } else {
            Rect Q = R;
            Q.y1 += L.dil * (L.k - 1); Q.x1 += L.dil * (L.k - 1);
            if (dims == 3) Q.z1 += L.dil * (L.k - 1);
            need[L.src].unite(clip(Q, L.src));
        }
    }
    if (g_trace_host)
        for (int i = 0; i < nl; ++i) {
            const tpz_layer& L = m->layers[i].L;
            const Rect& r = need[L.dst];
            fprintf(stderr, "[tpz host] need_regions: layer %d op %d -> slot %d: z [%d, %d) of %d, y [%d, %d) of %d, x [%d, %d) of %d\n", i,
                    (int)L.op, L.dst, r.z0, r.z1, Ds[L.dst], r.y0, r.y1, Hs[L.dst], r.x0, r.x1, Ws[L.dst]);
        }
    return need;
}

static int run_program(tpz_model* m, std::vector<Slot>& slots, float* d_out, const float* d_nrm, bool split = false,
                       const Rect* keep = nullptr) {
    tpz_ctx* ctx = m->ctx;
    const int nl = (int)m->layers.size();
    slots.resize(std::max<size_t>(slots.size(), (size_t)m->n_slots));
    std::vector<Rect> need;
    if (keep && slots[0].set) need = need_regions(m, slots[0].D, slots[0].H, slots[0].W, *keep, split);
    int rc = 0;
    for (int i = 0; i < nl && rc == 0; ++i) {
        const LayerRT& rt = m->layers[i];
        const tpz_layer& L = rt.L;
        const Slot& s1 = slots[L.src];
        if (!s1.set) { rc = fail(ctx, "layer %d reads unset slot %d", i, L.src); break; }
        Slot& dst = slots[L.dst];
        if (L.op == TPZ_OP_CONV && split && rt.folded_into >= 0 && m->layers[rt.folded_into].ks_fold) {
            // a 1x1 projection folded into the conv that adds it (prepare_split): nothing to run, its slot stays unset
        } else if (L.op == TPZ_OP_CONV) {
            const bool fold_here = split && rt.ks_fold && rt.fold_src >= 0 && slots[rt.fold_src].set;
            const Slot* s2 = L.src2 >= 0 ? &slots[L.src2] : nullptr;
            const Slot* sres = (L.res >= 0 && !fold_here) ? &slots[L.res] : nullptr;
            if ((s2 && !s2->set) || (sres && !sres->set)) { rc = fail(ctx, "layer %d reads an unset slot", i); break; }
            const Slot& geo = s2 ? *s2 : s1;
            if (s1.C + (s2 ? s2->C : 0) != L.cin) {
                rc = fail(ctx, "layer %d: cin %d != channels of its sources (%d)", i, L.cin, s1.C + (s2 ? s2->C : 0));
                break;
            }
            const int span = L.dil * (L.k - 1);
            const int Do = L.dims == 3 ? geo.D + 2 * L.pad - span : 1;
            const int Ho = geo.H + 2 * L.pad - span, Wo = geo.W + 2 * L.pad - span;
            if (Do < 1 || Ho < 1 || Wo < 1) { rc = fail(ctx, "layer %d: input %dx%dx%d too small", i, geo.D, geo.H, geo.W); break; }
            const int Co = L.head ? 1 : L.cout;
            if (sres && (sres->H - 2 * L.res_crop != Ho || sres->W - 2 * L.res_crop != Wo || sres->C != L.cout)) {
                rc = fail(ctx, "layer %d: residual geometry mismatch", i);
                break;
            }
            // which kernels run the layer: the 2xf16 per-parity twin, a 2xf16 kernel, or the fp32 path
            const bool exact2x = s2 && s2->H == 2 * s1.H && s2->W == 2 * s1.W && (L.dims == 2 || s2->D == 2 * s1.D);
            const bool use_sphase = split && rt.sphase.valid && exact2x;
            const bool use_split = split && rt.ks && !use_sphase;      // (fold_here implies it)
            const bool use_stem = split && rt.ks_stem && !s1.split;
            const bool use_last = split && rt.ks_last;
            const bool stem_split = split && !use_sphase && !use_split && !use_stem && rt.ki_stem_split;
            const bool split_dst = use_sphase || stem_split || use_stem || (use_split && !L.head && rt.ks->epi != EPI_PLAIN_F32);
            // the max-pool that follows is applied in this conv's epilogue: the slot receives the pooled tensor
            const bool fuse_pool = rt.ks_pool && (use_stem || (use_split && !s2)) && i + 1 < nl;
            const int Hd = fuse_pool ? Ho / 2 : Ho, Wd = fuse_pool ? Wo / 2 : Wo;
            // split tensors take the bytes of fp32 with the channels rounded up to whole 8-channel cells
            const size_t c_alloc = split_dst ? split_cells(Co) * 8 : (size_t)Co;
            float* p = (i == nl - 1) ? d_out : (float*)pool_alloc(ctx, c_alloc * Do * Hd * Wd * sizeof(float));
            if (!p) { rc = fail(ctx, "out of device memory (layer %d)", i); break; }
            if (fuse_pool && (Hd < 1 || Wd < 1)) { rc = fail(ctx, "layer %d: input too small to pool", i + 1); break; }
            set_dense(dst, p, Co, Do, Hd, Wd);
            dst.need = need.empty() ? Rect() : need[L.dst];
            dst.split = split_dst;
            dst.pooled = fuse_pool;
            dst.alt = nullptr;
            dst.owned = (i != nl - 1);
            if (split_dst && i == nl - 1) { rc = fail(ctx, "layer %d: the result must leave as fp32", i); break; }
            // sources in the format the chosen kernels read (converted once if the producer wrote the other one)
            const bool want1 = use_sphase || use_split || use_last;
            const bool want2 = use_sphase ? !rt.sphase.ki_skip_stem : use_split;
            Slot v1 = s1, v2, vres;
            v1.p = slot_as(ctx, slots[L.src], want1);
            v1.split = want1;
            if (s2) { v2 = *s2; v2.p = slot_as(ctx, slots[L.src2], want2); v2.split = want2; }
            if (sres) { vres = *sres; vres.p = slot_as(ctx, slots[L.res], use_split && !use_last); vres.split = use_split && !use_last; }
            if (!v1.p || (s2 && !v2.p) || (sres && !vres.p)) { rc = fail(ctx, "layer %d: tensor format conversion failed", i); break; }
            // slot 0 arrives already normalised (denoise_region); only the last layer un-normalises
            if (use_stem) rc = run_stem_split(ctx, rt, v1, dst, fuse_pool);
            else if (use_last) rc = run_last_split(ctx, rt, v1, dst, d_nrm, (d_nrm && i == nl - 1) ? 1 : 0, sres ? &vres : nullptr);
            else if (use_sphase) rc = run_conv_split_phases(ctx, rt, v1, v2, dst);
            else if (use_split && fold_here) {
                Slot vf = slots[rt.fold_src];
                vf.p = slot_as(ctx, slots[rt.fold_src], true);
                vf.split = true;
                if (!vf.p) { rc = fail(ctx, "layer %d: tensor format conversion failed", i); break; }
                rc = run_conv_split(ctx, rt, v1, nullptr, dst, nullptr, false, &vf);
            }
            else if (use_split) rc = run_conv_split(ctx, rt, v1, sres ? &vres : nullptr, dst, s2 ? &v2 : nullptr, fuse_pool);
            else rc = run_conv(ctx, rt, v1, s2 ? &v2 : nullptr, sres ? &vres : nullptr, dst, d_nrm,
                               (d_nrm && i == nl - 1) ? 1 : 0, stem_split);
        } else if (L.op == TPZ_OP_MAXPOOL2 && s1.pooled && L.dims == 3) {
            // pooled in-plane by the producing conv: the z pairs remain
            const int Do = s1.D / 2;
            if (Do < 1) { rc = fail(ctx, "layer %d: input too small to pool", i); break; }
            float* p = (i == nl - 1) ? d_out : (float*)pool_alloc(ctx, split_cells(s1.C) * 8 * (size_t)Do * s1.H * s1.W * sizeof(float));
            if (!p) { rc = fail(ctx, "out of device memory (layer %d)", i); break; }
            const Slot src = s1;
            set_dense(dst, p, src.C, Do, src.H, src.W);
            dst.split = true;
            dst.alt = nullptr;
            dst.owned = (i != nl - 1);
            hipError_t e;
            {
                const float* sp_ = src.p; float* dp_ = dst.p;
                const int C = src.C, Dd = src.D, Hh = src.H, Ww = src.W;
                e = enqueue(ctx, [=](hipStream_t st) { return launch_maxpoolz_split(sp_, dp_, C, Dd, Hh, Ww, st); });
            }
            if (e != hipSuccess) rc = fail(ctx, "maxpool launch failed: %s", hipGetErrorString(e));
        } else if (L.op == TPZ_OP_MAXPOOL2 && s1.pooled) {
            // already pooled by the producing conv: the slot changes hands
            dst = s1;
            dst.need = need.empty() ? Rect() : need[L.dst];
            dst.pooled = false;
            slots[L.src].owned = false;
            slots[L.src].alt = nullptr;
        } else if (L.op == TPZ_OP_MAXPOOL2) {
            if (s1.pitch != s1.W || s1.ps != (long long)s1.H * s1.W) { rc = fail(ctx, "maxpool needs a dense input"); break; }
            const int Do = L.dims == 3 ? s1.D / 2 : 1, Ho = s1.H / 2, Wo = s1.W / 2;
            if (Do < 1 || Ho < 1 || Wo < 1) { rc = fail(ctx, "layer %d: input too small to pool", i); break; }
            const bool sp = s1.split && i != nl - 1;          // pooled in the format the source has
            const float* src_p = s1.p;
            if (s1.split && !sp) { src_p = slot_as(ctx, slots[L.src], false); if (!src_p) { rc = fail(ctx, "conversion failed"); break; } }
            const size_t c_alloc = sp ? split_cells(s1.C) * 8 : (size_t)s1.C;
            float* p = (i == nl - 1) ? d_out : (float*)pool_alloc(ctx, c_alloc * Do * Ho * Wo * sizeof(float));
            if (!p) { rc = fail(ctx, "out of device memory (layer %d)", i); break; }
            const int Cs = s1.C, Ds = s1.D, Hs = s1.H, Ws = s1.W;
            set_dense(dst, p, Cs, Do, Ho, Wo);
            dst.split = sp;
            dst.alt = nullptr;
            dst.owned = (i != nl - 1);
            hipError_t e;
            {
                float* dp_ = dst.p;
                const int dims = L.dims;
                e = enqueue(ctx, [=](hipStream_t st) {
                    return sp ? launch_maxpool2_split(src_p, dp_, Cs, Ds, Hs, Ws, dims, st) : launch_maxpool2(src_p, dp_, Cs, Ds, Hs, Ws, dims, st);
                });
            }
            if (e != hipSuccess) rc = fail(ctx, "maxpool launch failed: %s", hipGetErrorString(e));
        } else if (L.op == TPZ_OP_MAXPOOL) {
            if (s1.pitch != s1.W || s1.ps != (long long)s1.H * s1.W) { rc = fail(ctx, "maxpool needs a dense input"); break; }
            const int span = L.dil * (L.k - 1);
            const int Do = L.dims == 3 ? s1.D - span : 1, Ho = s1.H - span, Wo = s1.W - span;
            if (Do < 1 || Ho < 1 || Wo < 1) { rc = fail(ctx, "layer %d: input too small to pool", i); break; }
            const bool sp = s1.split && i != nl - 1;          // pooled in the format the source has
            const float* src_p = s1.p;
            if (s1.split && !sp) { src_p = slot_as(ctx, slots[L.src], false); if (!src_p) { rc = fail(ctx, "conversion failed"); break; } }
            const size_t c_alloc = sp ? split_cells(s1.C) * 8 : (size_t)s1.C;
            float* p = (i == nl - 1) ? d_out : (float*)pool_alloc(ctx, c_alloc * Do * Ho * Wo * sizeof(float));
            if (!p) { rc = fail(ctx, "out of device memory (layer %d)", i); break; }
            const int Cs = s1.C, Ds = s1.D, Hs = s1.H, Ws = s1.W;
            set_dense(dst, p, Cs, Do, Ho, Wo);
            dst.split = sp;
            dst.alt = nullptr;
            dst.owned = (i != nl - 1);
            hipError_t e;
            {
                float* dp_ = dst.p;
                const int k = L.k, dil = L.dil, dims = L.dims;
                e = enqueue(ctx, [=](hipStream_t st) { return launch_maxpoolk(src_p, dp_, Cs, Ds, Hs, Ws, k, dil, dims, sp, st); });
            }
            if (e != hipSuccess) rc = fail(ctx, "maxpool launch failed: %s", hipGetErrorString(e));
        } else {
            rc = fail(ctx, "layer %d: unknown op %d", i, L.op);
        }
        // release intermediates whose last reader was this layer
        for (int s = 0; s < m->n_slots; ++s)
            if (slots[s].set && m->last_use[s] == i) {
                if (slots[s].owned) { pool_release(ctx, slots[s].p); slots[s].owned = false; }
                if (slots[s].alt) { pool_release(ctx, slots[s].alt); slots[s].alt = nullptr; }
            }
    }
    for (auto& s : slots) {
        if (s.owned) { pool_release(ctx, s.p); s.owned = false; }
        if (s.alt) { pool_release(ctx, s.alt); s.alt = nullptr; }
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* tpz_version(void) { return "topaz_hip 0.1 (gfx950)"; }

const char* tpz_last_error(tpz_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int tpz_ctx_create(int device_id, tpz_ctx** out) {
    if (!out) return fail(nullptr, "tpz_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, "tpz_ctx_create: no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, "tpz_ctx_create: device %d of %d", device_id, ndev);
    tpz_ctx* ctx = new tpz_ctx();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return fail(nullptr, "hipSetDevice(%d) failed", device_id); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            delete ctx;
            return fail(nullptr, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                        prop.gcnArchName);
        }
        if (prop.multiProcessorCount > 0) ctx->n_cus = prop.multiProcessorCount;
    }
    if (const char* e = getenv("TPZ_LANES")) {
        const int n = atoi(e);
        if (n >= 2 && n <= N_LANES) ctx->n_lanes = n;
    }
    if (hipStreamCreate(&ctx->own_stream) != hipSuccess) { delete ctx; return fail(nullptr, "hipStreamCreate failed"); }
    ctx->stream = ctx->own_stream;
    if (hipMalloc((void**)&ctx->d_part, 2 * PART_BLOCKS * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_nrm, 4 * NRM_RING * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_counters, NMS_COUNTERS * sizeof(unsigned int)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_flag, 16) != hipSuccess || hipHostMalloc((void**)&ctx->h_flag, 16) != hipSuccess ||
        hipMalloc((void**)&ctx->d_absmax, 1024) != hipSuccess || hipMemset(ctx->d_absmax, 0, 1024) != hipSuccess ||
        hipMalloc((void**)&ctx->d_zeros, 256) != hipSuccess || hipMemset(ctx->d_zeros, 0, 256) != hipSuccess) {
        delete ctx;
        return fail(nullptr, "tpz_ctx_create: hipMalloc failed");
    }
    *out = ctx;
    return 0;
}

void tpz_ctx_destroy(tpz_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipDeviceSynchronize();
    if (ctx->io_stage) tpz_stage_free(ctx->io_stage);
    for (auto& b : ctx->pool) (void)hipFree(b.p);
    for (auto& lane_pools : ctx->rec_pools)
        for (auto& rp : lane_pools)
            for (auto& b : rp) (void)hipFree(b.p);
    for (auto& ln : ctx->lanes) {
        for (auto& b : ln.pool) (void)hipFree(b.p);
        if (ln.d_part) (void)hipFree(ln.d_part);
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    if (ctx->lanes_fork) (void)hipEventDestroy(ctx->lanes_fork);
    (void)hipFree(ctx->d_part);
    (void)hipFree(ctx->d_nrm);
    (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->d_zeros);
    (void)hipFree(ctx->d_flag);
    (void)hipFree(ctx->d_absmax);
    (void)hipHostFree(ctx->h_flag);
    for (auto& e : ctx->split_plans) (void)hipFree(e.d);
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int tpz_ctx_set_stream(tpz_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    (void)hipStreamSynchronize(ctx->stream);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

int tpz_ctx_sync(tpz_ctx* ctx) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
                      const std::vector<int>& preset_chan, tpz_model** out);

int tpz_model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
                   tpz_model** out) {
    return model_load(ctx, layers, n_layers, h_blob, n_floats, {1}, out);    // slot 0 = the 1-channel input
}

// Moves every bias-like vector of the model (each chan_pad-ed) into ONE device array and allocates a second one of the same size:
// a range-scaled pass (tpz_model_forward) writes 2^-s * arena there with one small kernel and reads its biases `bias_shift`
// floats further on.
static int build_bias_arena(tpz_ctx* ctx, tpz_model* m) {
    std::vector<std::pair<float**, size_t>> vecs;
    for (LayerRT& rt : m->layers) {
        if (rt.L.op != TPZ_OP_CONV) continue;
        const size_t n = chan_pad((size_t)rt.L.cout);
        if (rt.d_bias) vecs.push_back({&rt.d_bias, n});
        if (rt.d_post_shift) vecs.push_back({&rt.d_post_shift, n});
        if (rt.d_bias_fold) vecs.push_back({&rt.d_bias_fold, n});
    }
    size_t total = 0;
    for (auto& v : vecs) total += v.second;
    if (total == 0) return 0;
    float *arena = nullptr, *scaled = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&arena, total * sizeof(float)));
    m->dev_allocs.push_back(arena);
    HIPCHK(ctx, hipMalloc((void**)&scaled, total * sizeof(float)));
    m->dev_allocs.push_back(scaled);
    size_t off = 0;
    for (auto& v : vecs) {
        HIPCHK(ctx, hipMemcpy(arena + off, *v.first, v.second * sizeof(float), hipMemcpyDeviceToDevice));
        *v.first = arena + off;              // (the vector's first home stays in dev_allocs and is freed with the model)
        off += v.second;
    }
    HIPCHK(ctx, hipMemcpy(scaled, arena, total * sizeof(float), hipMemcpyDeviceToDevice));
    m->d_bias_arena = arena; m->d_bias_scaled = scaled; m->n_bias_arena = total;
    return 0;
}

// The 2xf16 kernels address channels in 8-channel cells and walk their K loop in chunks of CC = 2 cells; a two-source layer
// (fused upsample + concat, denoising/models.py:140-171) needs its first source to fill whole chunks.  A user-trained width
// that is not a multiple of 16 (`UDenoiseNet2(nf=12)`: 24 -> 24 over sources of 12 + 12, 25 -> 64 over 24 + 1) therefore fell
// to the fp32-MFMA kernels, 3 - 5x slower.  widen_program rewrites such a program with every intermediate tensor ZERO-PADDED to
// the next multiple of 16 channels: padded output channels get zero weights / bias / affine / head weights (they come out as
// exactly 0 through any activation with f(0) = 0), padded input channels zero weight columns, the channels of a second
// source move up behind the padded first one.  Every real product and every real sum stays what it was (zeros added in
// fp32): same arithmetic on the same values.  The 1-channel input, 1-output-channel convs, the fused head's single channel
// and the network's last layer keep their widths.  Returns false when nothing needs padding.
static bool widen_program(const tpz_layer* layers, int n_layers, const float* blob, size_t n_floats, std::vector<tpz_layer>& out_l,
                          std::vector<float>& out_b) {
    int max_slot = 0;
    for (int i = 0; i < n_layers; ++i)
        max_slot = std::max(max_slot, std::max(std::max(layers[i].src, layers[i].src2), std::max(layers[i].dst, layers[i].res)));
    std::vector<int> chan(max_slot + 1, 0), pch(max_slot + 1, 0);
    chan[0] = pch[0] = 1;
    auto pad16 = [](int c) { return c <= 1 ? c : (c + 15) / 16 * 16; };
    bool any = false;
    out_l.assign(layers, layers + n_layers);
    out_b.assign(blob, blob + n_floats);
    for (int i = 0; i < n_layers; ++i) {
        tpz_layer& L = out_l[i];
        if (L.src < 0 || L.src > max_slot || L.dst <= 0) return false;              // (model_load reports the bad program)
        const int c1 = chan[L.src], c2 = L.src2 >= 0 ? chan[L.src2] : 0;
        const int p1 = pch[L.src], p2 = L.src2 >= 0 ? pch[L.src2] : 0;
        if (L.op != TPZ_OP_CONV) { chan[L.dst] = c1; pch[L.dst] = p1; continue; }
        if (L.cin != c1 + c2 || L.w_off < 0) return false;
        const size_t taps = L.dims == 3 ? (size_t)L.k * L.k * L.k : (size_t)L.k * L.k;
        if ((size_t)L.w_off + (size_t)L.cout * L.cin * taps > n_floats) return false;
        const bool last = i == n_layers - 1;
        const int pco = (last || L.cout == 1) ? L.cout : pad16(L.cout);
        chan[L.dst] = L.head ? 1 : L.cout;
        pch[L.dst] = L.head ? 1 : pco;
        if (pco == L.cout && p1 == c1 && p2 == c2) continue;
        any = true;
        const int pci = p1 + p2;
        const size_t w_new = out_b.size();
        out_b.resize(w_new + (size_t)pco * pci * taps, 0.f);
        for (int co = 0; co < L.cout; ++co) {
            memcpy(&out_b[w_new + ((size_t)co * pci) * taps], blob + L.w_off + ((size_t)co * L.cin) * taps, (size_t)c1 * taps * sizeof(float));
            if (c2 > 0)
                memcpy(&out_b[w_new + ((size_t)co * pci + p1) * taps], blob + L.w_off + ((size_t)co * L.cin + c1) * taps,
                       (size_t)c2 * taps * sizeof(float));
        }
        auto widen_vec = [&](int64_t& off) {
            if (off < 0 || pco == L.cout) return;
            const size_t o = out_b.size();
            out_b.resize(o + pco, 0.f);
            memcpy(&out_b[o], blob + off, (size_t)L.cout * sizeof(float));
            off = (int64_t)o;
        };
        widen_vec(L.b_off); widen_vec(L.post_scale_off); widen_vec(L.post_shift_off);
        if (L.head) widen_vec(L.head_w_off);
        L.w_off = (int64_t)w_new;
        L.cin = pci;
        L.cout = pco;
    }
    return any;
}

static int model_load_one(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
                          const std::vector<int>& preset_chan, tpz_model** out);

// preset_chan: channels of the externally provided slots (slot 0, and tpz_conv's extra sources)
static int model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
                      const std::vector<int>& preset_chan, tpz_model** out) {
    if (model_load_one(ctx, layers, n_layers, h_blob, n_floats, preset_chan, out)) return 1;
    tpz_model* m = *out;
    static const bool no_widen = getenv("TPZ_NO_WIDEN") != nullptr;         // A/B switch
    if (preset_chan.size() != 1 || no_widen || ctx->exact || m->n_conv_split == m->n_conv) return 0;
    // some layer has no 2xf16 kernel at the widths as given: try the zero-padded program, keep whichever covers more layers
    std::vector<tpz_layer> wl;
    std::vector<float> wb;
    if (!widen_program(layers, n_layers, h_blob, n_floats, wl, wb)) return 0;
    tpz_model* mw = nullptr;
    if (model_load_one(ctx, wl.data(), n_layers, wb.data(), wb.size(), preset_chan, &mw)) return 0;    // (keep the plain one)
    if (mw->n_conv - mw->n_conv_split < m->n_conv - m->n_conv_split) {
        mw->widened = true;
        tpz_model_free(m);
        *out = mw;
    } else {
        tpz_model_free(mw);
    }
    return 0;
}

static int model_load_one(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
                          const std::vector<int>& preset_chan, tpz_model** out) {
    if (!ctx || !layers || !out || n_layers < 1) return fail(ctx, "tpz_model_load: bad arguments");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    tpz_model* m = new tpz_model();
    m->ctx = ctx;
    int max_slot = 0;
    for (int i = 0; i < n_layers; ++i) {
        const tpz_layer& L = layers[i];
        max_slot = std::max(max_slot, std::max(std::max(L.src, L.src2), std::max(L.dst, L.res)));
        if (L.src < 0 || L.dst <= 0) { tpz_model_free(m); return fail(ctx, "layer %d: bad slot ids", i); }
    }
    m->n_slots = max_slot + 1;
    m->last_use.assign(m->n_slots, -1);
    m->layers.resize(n_layers);
    std::vector<int> chan(m->n_slots, 0);       // channels of every slot
    for (size_t i = 0; i < preset_chan.size() && i < chan.size(); ++i) chan[i] = preset_chan[i];
    for (int i = 0; i < n_layers; ++i) {
        const tpz_layer& L = layers[i];
        m->last_use[L.src] = i;
        if (L.src2 >= 0) m->last_use[L.src2] = i;
        if (L.res >= 0) m->last_use[L.res] = i;
        const int c1 = chan[L.src], c2 = L.src2 >= 0 ? chan[L.src2] : 0;
        if (prepare_layer(ctx, m, L, h_blob, n_floats, m->layers[i], c1, c2)) { tpz_model_free(m); return 1; }
        chan[L.dst] = L.op == TPZ_OP_CONV ? (L.head ? 1 : L.cout) : c1;
    }
    if (preset_chan.size() == 1 && prepare_split(ctx, m, h_blob)) { tpz_model_free(m); return 1; }
    if (preset_chan.size() == 1 && build_bias_arena(ctx, m)) { tpz_model_free(m); return 1; }
    *out = m;
    return 0;
}

void tpz_model_free(tpz_model* m) {
    if (!m) return;
    if (m->ctx) (void)hipStreamSynchronize(m->ctx->stream);
    for (void* p : m->dev_allocs) (void)hipFree(p);
    delete m;
}

int tpz_model_out_shape(tpz_model* m, int D, int H, int W, int* Do, int* Ho, int* Wo) {
    if (!m) return fail(nullptr, "model is NULL");
    struct S { int C, D, H, W; };
    std::vector<S> s(m->n_slots, S{0, 0, 0, 0});
    s[0] = {1, D, H, W};
    for (auto& rt : m->layers) {
        const tpz_layer& L = rt.L;
        const S& g = L.src2 >= 0 ? s[L.src2] : s[L.src];
        if (L.op == TPZ_OP_CONV) {
            const int span = L.dil * (L.k - 1);
            s[L.dst] = {L.head ? 1 : L.cout, L.dims == 3 ? g.D + 2 * L.pad - span : 1, g.H + 2 * L.pad - span,
                        g.W + 2 * L.pad - span};
        } else if (L.op == TPZ_OP_MAXPOOL) {
            const int span = L.dil * (L.k - 1);
            s[L.dst] = {g.C, L.dims == 3 ? g.D - span : 1, g.H - span, g.W - span};
        } else {
            s[L.dst] = {g.C, L.dims == 3 ? g.D / 2 : 1, g.H / 2, g.W / 2};
        }
    }
    const S& o = s[m->layers.back().L.dst];
    if (Do) *Do = o.D;
    if (Ho) *Ho = o.H;
    if (Wo) *Wo = o.W;
    return 0;
}

int tpz_model_out_channels(tpz_model* m, int* C) {
    if (!m || !C) return fail(nullptr, "tpz_model_out_channels: NULL argument");
    const tpz_layer& L = m->layers.back().L;
    if (L.op == TPZ_OP_CONV) { *C = L.head ? 1 : L.cout; return 0; }
    // pooling keeps the channels of its source conv
    for (int i = (int)m->layers.size() - 1; i >= 0; --i)
        if (m->layers[i].L.op == TPZ_OP_CONV) { *C = m->layers[i].L.head ? 1 : m->layers[i].L.cout; return 0; }
    *C = 1;
    return 0;
}

// The plane-stacked 3-D kernels address a whole split tensor half with 32-bit byte offsets (conv_split.h fetch): a volume whose
// widest activation exceeds 4 GiB per half stays on the fp32 kernels.  (No tensor of these networks is larger than the input
// in voxels: 'same' or valid convolutions, pools.)
static bool split_volume_fits(const tpz_model* m, int D, int H, int W) {
    if (D <= 1) return true;
    size_t cmax = 1;
    for (const LayerRT& rt : m->layers)
        if (rt.L.op == TPZ_OP_CONV) cmax = std::max(cmax, (size_t)std::max(rt.L.cin, rt.L.cout));
    return split_cells((int)cmax) * (size_t)D * H * W * 16 < ((size_t)1 << 32) - 16;
}

// One image through the program -- whole, or, a 2-D image above the tiling limit through a size-preserving network, TILE by
// tile: `topaz extract` scores any image that fits memory (topaz/extract.py:247-249), while the kernels address a chunk of cells
// with 32-bit byte offsets (< 4 GiB: ~11 500^2 pixels) and a whole-image pass of a large detector frame holds every activation at
// full size.  The filled network is translation-equivariant with a finite receptive field: a tile's outputs are those of the
// network run on the tile grown by that halo (clipped at the image borders, where the layers' own zero padding applies as it
// does on the whole image), minus the halo ring.  Every kept logit is computed by the same instructions on the same operands as
// in a whole-image pass: bit-identical (tests/test_gpu_scoring.py::test_internal_tiling_is_bit_identical).
static int model_halo(const tpz_model* m) {
    int h = 0;
    for (const LayerRT& rt : m->layers) {
        const tpz_layer& L = rt.L;
        if (L.op == TPZ_OP_CONV) h += std::max(L.pad, L.dil * (L.k - 1) - L.pad);        // (an upper bound: every layer counted)
        else if (L.op == TPZ_OP_MAXPOOL) h += L.dil * (L.k - 1);
        else return -1;                                                                   // (pooling by 2: not equivariant)
    }
    return (h + 1) & ~1;
}

static int run_image(tpz_model* m, float* x, int D, int H, int W, float* out, int Co, int Do, int Ho, int Wo, bool split) {
    tpz_ctx* ctx = m->ctx;
    const int halo = (D == 1 && Ho == H && Wo == W) ? model_halo(m) : -1;
    if (halo < 0 || (long long)H * W <= ctx->tile_limit_px) {
        std::vector<Slot> slots(m->n_slots);
        set_dense(slots[0], x, 1, D, H, W);
        return run_program(m, slots, out, nullptr, split);
    }
    const int T = std::max(16, ctx->tile_size);
    for (int ty = 0; ty < H; ty += T)
        for (int tx = 0; tx < W; tx += T) {
            const int y0 = std::max(0, ty - halo), y1 = std::min(H, ty + T + halo);
            const int x0 = std::max(0, tx - halo), x1 = std::min(W, tx + T + halo);
            const int wh = y1 - y0, ww = x1 - x0, th = std::min(T, H - ty), tw = std::min(T, W - tx);
            float* xt = (float*)pool_alloc(ctx, (size_t)wh * ww * sizeof(float));
            float* ot = (float*)pool_alloc(ctx, (size_t)Co * wh * ww * sizeof(float));
            int rc = (!xt || !ot) ? fail(ctx, "out of device memory") : 0;
            if (!rc) {
                const float* src = x + (size_t)y0 * W + x0;
                const hipError_t e = enqueue(ctx, [=](hipStream_t st) { return launch_copy_box(src, 0, W, xt, 0, ww, 1, wh, ww, st); });
                if (e != hipSuccess) rc = fail(ctx, "copy_box failed: %s", hipGetErrorString(e));
            }
            if (!rc) {
                std::vector<Slot> slots(m->n_slots);
                set_dense(slots[0], xt, 1, 1, wh, ww);
                rc = run_program(m, slots, ot, nullptr, split);
            }
            if (!rc) {
                const float* src = ot + (size_t)(ty - y0) * ww + (tx - x0);
                float* dst = out + (size_t)ty * W + tx;
                const hipError_t e = enqueue(ctx, [=](hipStream_t st) {
                    return launch_copy_box(src, (long long)wh * ww, ww, dst, (long long)H * W, W, Co, th, tw, st);
                });
                if (e != hipSuccess) rc = fail(ctx, "copy_box failed: %s", hipGetErrorString(e));
            }
            if (xt) pool_release(ctx, xt);
            if (ot) pool_release(ctx, ot);
            if (rc) return rc;
        }
    return 0;
}

int tpz_model_forward(tpz_model* m, const float* d_in, int n, int D, int H, int W, float* d_out) {
    if (!m || !d_in || !d_out) return fail(m ? m->ctx : nullptr, "tpz_model_forward: NULL argument");
    tpz_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int Do, Ho, Wo;
    tpz_model_out_shape(m, D, H, W, &Do, &Ho, &Wo);
    if (Do < 1 || Ho < 1 || Wo < 1) return fail(ctx, "input %dx%dx%d too small for this model", D, H, W);
    int Co = 1;
    tpz_model_out_channels(m, &Co);
    for (int b = 0; b < n; ++b) {
        float* out_b = d_out + (size_t)b * Co * Do * Ho * Wo;
        bool done = false;
        if (m->split_ok && !ctx->exact && split_volume_fits(m, D, H, W)) {
            // 2xf16 path; an activation beyond the f16 range (flag) sends this image to the fp32 kernels instead
            HIPCHK(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(unsigned), ctx->stream));
            float* x_b = const_cast<float*>(d_in) + (size_t)b * D * H * W;
            // RANGE SCALING (scoring networks = programs ending in the linear head): `topaz extract` does not normalise its
            // input (extract.py:234-249), and a raw-count micrograph would leave the f16 range in the stem.  The network is
            // positively homogeneous in (input, biases): it runs on x * 2^-s with its biases scaled alike and the logits are
            // multiplied back -- exact, powers of two; s follows the BULK of the image (its 99.9 % quantile of |x| -> ~8; s = 0 for
            // a normalised image), so outliers cannot starve the rest of precision (kernels_misc.hip launch_range_fit).
            const bool scaled = ctx->range_scaling && m->layers.back().L.op == TPZ_OP_CONV && m->layers.back().L.head &&
                                m->d_bias_scaled != nullptr;
            float *xs = nullptr, *rng = nullptr;
            if (scaled) {
                const size_t n_in = (size_t)D * H * W;
                rng = next_nrm(ctx);
                xs = (float*)pool_alloc(ctx, n_in * sizeof(float));
                if (!xs) return fail(ctx, "out of device memory");
                hipError_t e = enqueue(ctx, [=](hipStream_t st) {
                    return launch_range_fit(x_b, n_in, ctx->d_absmax, rng, m->d_bias_arena, m->d_bias_scaled, m->n_bias_arena, st);
                });
                if (e == hipSuccess) e = enqueue(ctx, [=](hipStream_t st) { return launch_affine_dev(x_b, D, H, W, (long long)H * W, W, rng, xs, st); });
                if (e != hipSuccess) { pool_release(ctx, xs); return fail(ctx, "range scaling failed: %s", hipGetErrorString(e)); }
                x_b = xs;
                ctx->bias_shift = m->d_bias_scaled - m->d_bias_arena;
                ctx->scaled_pass = true;
            }
            const int rc = run_image(m, x_b, D, H, W, out_b, Co, Do, Ho, Wo, true);
            ctx->bias_shift = 0;
            ctx->scaled_pass = false;
            if (xs) pool_release(ctx, xs);
            if (rc) return 1;
            if (scaled) {
                const size_t n_out = (size_t)Co * Do * Ho * Wo;
                const float hb = m->layers.back().head_b;
                HIPCHK(ctx, enqueue(ctx, [=](hipStream_t st) { return launch_unscale(out_b, n_out, rng, hb, st); }));
            }
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_flag, ctx->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            done = (*ctx->h_flag == 0);
            if (done) ++m->n_split; else ++m->n_fallback;
        }
        if (!done && run_image(m, const_cast<float*>(d_in) + (size_t)b * D * H * W, D, H, W, out_b, Co, Do, Ho, Wo, false)) return 1;
    }
    return 0;
}

int tpz_ctx_set_lanes(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (on < 0 || on > N_LANES) return fail(ctx, "tpz_ctx_set_lanes: 0 (off), 1 (on, two lanes) or a lane count up to %d", (int)N_LANES);
    ctx->lanes_enabled = on != 0;
    if (on >= 1) ctx->n_lanes = on == 1 ? 2 : on;
    return 0;
}

int tpz_ctx_set_batch(tpz_ctx* ctx, int n) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    if (n < 0 || n > (int)SPLIT_MULTI_MAX) return fail(ctx, "tpz_ctx_set_batch: 0 (off) or up to %d images per launch", (int)SPLIT_MULTI_MAX);
    ctx->batch = n == 1 ? 0 : n;
    return 0;
}
int tpz_ctx_set_batch_memory(tpz_ctx* ctx, long long bytes) {
    if (!ctx || bytes < 0) return fail(ctx, "tpz_ctx_set_batch_memory: bytes >= 0 (0: 90 %% of the free device memory)");
    ctx->batch_mem = bytes;
    return 0;
}
long long tpz_prof_launches(tpz_ctx* ctx) { return ctx ? ctx->n_launches : 0; }

int tpz_ctx_set_persist(tpz_ctx* ctx, int mode, int workgroups) {
    if (!ctx || mode < 0 || mode > 2 || workgroups < 0) return fail(ctx, "tpz_ctx_set_persist: bad arguments");
    ctx->persist_mode = mode;
    ctx->persist_wgs = workgroups;
    return 0;
}
int tpz_ctx_set_roi(tpz_ctx* ctx, int on) {
    if (!ctx) return 1;
    ctx->roi_enabled = on != 0;
    return 0;
}

int tpz_ctx_set_tiling(tpz_ctx* ctx, long long limit_px, int tile) {
    if (!ctx || limit_px < 1 || tile < 16) return fail(ctx, "tpz_ctx_set_tiling: limit_px >= 1, tile >= 16");
    ctx->tile_limit_px = limit_px;
    ctx->tile_size = tile;
    return 0;
}

int tpz_ctx_set_raster(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    ctx->raster = on != 0;
    return 0;
}

int tpz_ctx_set_rw(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    ctx->rw_enabled = on != 0;
    return 0;
}
int tpz_ctx_set_range(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    ctx->range_scaling = on != 0;
    return 0;
}

int tpz_ctx_set_exact(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    ctx->exact = (on != 0) || g_exact_fp32;
    return 0;
}

int tpz_model_split_stats(tpz_model* m, int* eligible, long long* split_runs, long long* fp32_reruns) {
    if (!m) return fail(nullptr, "model is NULL");
    if (eligible) *eligible = m->split_ok ? 1 : 0;
    if (split_runs) *split_runs = m->n_split;
    if (fp32_reruns) *fp32_reruns = m->n_fallback;
    return 0;
}

int tpz_model_split_layers(tpz_model* m, int* n_conv, int* n_split, char* off_path, int off_path_len) {
    if (!m) return fail(nullptr, "model is NULL");
    if (n_conv) *n_conv = m->n_conv;
    if (n_split) *n_split = m->n_conv_split;
    if (off_path && off_path_len > 0) snprintf(off_path, (size_t)off_path_len, "%s", m->off_path.c_str());
    return 0;
}

int tpz_conv_split_2d(tpz_ctx* ctx, const float* d_in, int cin, int H, int W, const float* h_w, const float* h_b,
                      int cout, int k, int dil, int pad, float slope, const float* d_res, int res_crop,
                      const float* h_post_scale, const float* h_post_shift, const float* h_head_w, float head_b,
                      float* d_out, int* overflow) {
    if (!ctx || !d_in || !h_w || !d_out) return fail(ctx, "tpz_conv_split_2d: NULL argument");
    if (slope > 1.f) return fail(ctx, "tpz_conv_split_2d: the 2xf16 epilogue applies max(v, slope * v): slope must be <= 1 (%g)", slope);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int epi = EPI_PLAIN;
    if (h_head_w) epi = EPI_HEAD;
    else if (d_res) epi = h_post_scale ? EPI_RES_POST : EPI_RES;
    LayerRT rt;
    tpz_layer& L = rt.L;
    memset(&L, 0, sizeof L);
    L.op = TPZ_OP_CONV; L.dims = 2; L.cin = cin; L.cout = cout; L.k = k; L.dil = dil; L.pad = pad; L.slope = slope;
    L.res = d_res ? 1 : -1; L.res_crop = res_crop; L.head = h_head_w ? 1 : 0; L.src2 = -1;
    rt.ks = pick_split(k, dil, cout, epi);
    if (!rt.ks) return fail(ctx, "tpz_conv_split_2d: no 2xf16 kernel for k=%d dil=%d cout=%d epi=%d", k, dil, cout, epi);
    const int span = dil * (k - 1);
    const int Ho = H + 2 * pad - span, Wo = W + 2 * pad - span;
    if (Ho < 1 || Wo < 1) return fail(ctx, "tpz_conv_split_2d: input too small");
    tpz_model tmp;
    tmp.ctx = ctx;
    rt.s_n_cog = (cout + rt.ks->MT - 1) / rt.ks->MT;
    rt.s_n_chunks = (int)((split_cells(cin) + rt.ks->CC - 1) / rt.ks->CC);
    std::vector<uint16_t> packed;
    std::vector<float> inv;
    pack_weights_split(*rt.ks, h_w, cout, cin, rt.s_n_cog, rt.s_n_chunks, packed, inv);
    float* d = nullptr;
    int rc = upload(ctx, &tmp, reinterpret_cast<const float*>(packed.data()), (packed.size() + 1) / 2, &d);
    rt.d_wsplit = d;
    if (!rc) rc = upload_chan(ctx, &tmp, inv.data(), inv.size(), &rt.d_wscale);
    if (!rc && h_b) rc = upload_chan(ctx, &tmp, h_b, cout, &rt.d_bias);
    if (!rc && h_post_scale) rc = upload_chan(ctx, &tmp, h_post_scale, cout, &rt.d_post_scale);
    if (!rc && h_post_shift) rc = upload_chan(ctx, &tmp, h_post_shift, cout, &rt.d_post_shift);
    if (!rc && h_head_w) { rc = upload_chan(ctx, &tmp, h_head_w, cout, &rt.d_head_w); rt.head_b = head_b; }
    Slot s1, sres, dst;
    float *x_s = nullptr, *r_s = nullptr, *y_s = nullptr;
    const int Hr = Ho + 2 * res_crop, Wr = Wo + 2 * res_crop;
    if (!rc) {
        x_s = (float*)pool_alloc(ctx, split_cells(cin) * 8 * (size_t)H * W * 4);
        y_s = (float*)pool_alloc(ctx, split_cells(cout) * 8 * (size_t)Ho * Wo * 4);
        if (d_res) r_s = (float*)pool_alloc(ctx, split_cells(cout) * 8 * (size_t)Hr * Wr * 4);
        if (!x_s || !y_s || (d_res && !r_s)) rc = fail(ctx, "out of device memory");
    }
    if (!rc) {
        (void)hipMemsetAsync(ctx->d_flag, 0, sizeof(unsigned), ctx->stream);
        (void)launch_to_split(d_in, x_s, cin, H, W, ctx->d_flag, ctx->stream);
        set_dense(s1, x_s, cin, 1, H, W); s1.split = true;
        if (d_res) { (void)launch_to_split(d_res, r_s, cout, Hr, Wr, ctx->d_flag, ctx->stream); set_dense(sres, r_s, cout, 1, Hr, Wr); sres.split = true; }
        set_dense(dst, L.head ? d_out : y_s, L.head ? 1 : cout, 1, Ho, Wo);
        rc = run_conv_split(ctx, rt, s1, d_res ? &sres : nullptr, dst);
        if (!rc && !L.head) (void)launch_from_split(y_s, d_out, cout, Ho, Wo, ctx->stream);
        (void)hipMemcpyAsync(ctx->h_flag, ctx->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, "tpz_conv_split_2d: kernel failed");
        if (overflow) *overflow = (int)*ctx->h_flag;
    }
    if (x_s) pool_release(ctx, x_s);
    if (y_s) pool_release(ctx, y_s);
    if (r_s) pool_release(ctx, r_s);
    for (void* p_ : tmp.dev_allocs) (void)hipFree(p_);
    return rc;
}

// ---- denoising ---------------------------------------------------------------------------------
// Denoise._denoise on a (strided) region: mean / unbiased std -> normalise -> network -> un-normalise.
// mode 1: plain; mode 2: the un-normalisation also applies the volume's std*y+mu with g = {mu, std}.
// How many patches / tiles a batched pass may hold at a time.  Every image of a batch lives on a workspace pool of its own and
// two batches are in flight (one per lane): at the CLI's defaults that is 16 x 4.5 GB for a tomogram and 16 x 2 GB for a
// micrograph -- nothing on 288 GB -- but a 384^3 tile is 8x that.  The batch is cut to what fits 90 % of the free device memory
// (+ what the pools already hold), counting for one image every tensor the program allocates (no reuse: an upper bound);
// below 2 the pass falls back to single patches on the lanes.
static int batch_that_fits(tpz_ctx* ctx, const tpz_model* m, int D, int H, int W) {
    if (ctx->batch < 2) return ctx->batch;
    struct S { int C, D, H, W; };
    std::vector<S> s(m->n_slots, S{0, 0, 0, 0});
    s[0] = {1, D, H, W};
    double per = 3.0 * 4.0 * D * H * W;                          // the tile, its normalised copy, the result
    for (auto& rt : m->layers) {
        const tpz_layer& L = rt.L;
        const S& g = L.src2 >= 0 ? s[L.src2] : s[L.src];
        if (L.op == TPZ_OP_CONV) {
            const int span = L.dil * (L.k - 1);
            s[L.dst] = {L.head ? 1 : L.cout, L.dims == 3 ? g.D + 2 * L.pad - span : 1, g.H + 2 * L.pad - span, g.W + 2 * L.pad - span};
            if (L.cin == 1 || L.cout == 1) per += 32.0 * ((L.k + 7) / 8) * (double)g.D * g.H * g.W;   // column-kernel copies
        } else if (L.op == TPZ_OP_MAXPOOL) {
            const int span = L.dil * (L.k - 1);
            s[L.dst] = {g.C, L.dims == 3 ? g.D - span : 1, g.H - span, g.W - span};
        } else {
            s[L.dst] = {g.C, L.dims == 3 ? g.D / 2 : 1, g.H / 2, g.W / 2};
        }
        const S& o = s[L.dst];
        if (o.D < 1 || o.H < 1 || o.W < 1) return ctx->batch;     // (the pass itself reports the bad geometry)
        per += 32.0 * split_cells(o.C) * (double)o.D * o.H * o.W;
    }
    double avail = (double)ctx->batch_mem;
    if (ctx->batch_mem <= 0) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return ctx->batch;
        size_t cached = 0;
        for (auto& lane_pools : ctx->rec_pools)
            for (auto& rp : lane_pools)
                for (auto& b : rp) cached += b.bytes;
        avail = 0.9 * ((double)free_b + (double)cached);
    }
    const int lanes = ctx->lanes_enabled ? ctx->n_lanes : 1;
    const int fit = (int)std::min<double>(ctx->batch, avail / (per * lanes));
    return fit >= 2 ? fit : 0;
}

static int denoise_region(tpz_model* m, const Slot& view, float* d_out_dense, int mode = 1,
                          const float* d_g = nullptr, bool split = false, const Rect* keep = nullptr) {
    tpz_ctx* ctx = m->ctx;
    float* nrm = next_nrm(ctx);
    const float* vp_ = view.p;
    const int vD = view.D, vH = view.H, vW = view.W, vpitch = view.pitch;
    const long long vps = view.ps;
    double* part = ctx->d_part;
    hipError_t e = enqueue(ctx, [=](hipStream_t st) {
        return launch_meanstd(vp_, vD, vH, vW, vps, vpitch, /*unbiased*/ 1, mode, d_g, part, PART_BLOCKS, nrm, st);
    });
    HIPCHK(ctx, e);
    // (x - mu)/std once, into a dense buffer every reader of slot 0 (first conv, dec1 concat) DMA-loads
    const size_t n = (size_t)view.D * view.H * view.W;
    float* xn = (float*)pool_alloc(ctx, n * sizeof(float));
    if (!xn) return fail(ctx, "out of device memory");
    e = enqueue(ctx, [=](hipStream_t st) { return launch_affine_dev(vp_, vD, vH, vW, vps, vpitch, nrm, xn, st); });
    if (e != hipSuccess) { pool_release(ctx, xn); return fail(ctx, "affine_dev failed: %s", hipGetErrorString(e)); }
    std::vector<Slot> slots(m->n_slots);
    set_dense(slots[0], xn, 1, view.D, view.H, view.W);
    const int rc = run_program(m, slots, d_out_dense, nrm, split, keep);
    pool_release(ctx, xn);
    return rc;
}

static int denoise_2d_pass(tpz_model* m, const float* d_in, int H, int W, int patch, int pad, float* d_out, bool split) {
    tpz_ctx* ctx = m->ctx;
    const int s = patch + pad;
    const bool use_patch = patch > 0 && (s < H || s < W);     // denoise.py:329-330
    if (!use_patch) {
        Slot v;
        set_dense(v, const_cast<float*>(d_in), 1, 1, H, W);
        return denoise_region(m, v, d_out, 1, nullptr, split);
    }
    // the patches are independent: on the 2xf16 path they run in batches (the same layer of `batch` patches in one launch:
    // rec_begin / rec_flush), otherwise alternating on the patch lanes
    const int batch = split ? batch_that_fits(ctx, m, 1, std::min(H, patch + 2 * pad), std::min(W, patch + 2 * pad)) : 0;
    const bool batched = batch >= 2;
    if (lanes_begin(ctx)) return 1;
    int rc_all = 0, n_patch = 0, slot = 0, slots_left = 0, n_batches = 0;
    for (int i = 0; i < H && !rc_all; i += patch)
        for (int j = 0; j < W && !rc_all; j += patch) {
            if (batched) {
                if (slots_left == 0) {
                    if (ctx->rec_on && rec_flush(ctx)) { rc_all = 1; break; }
                    rec_begin(ctx, n_batches++);
                    slots_left = batch;
                    slot = 0;
                }
                rec_select(ctx, slot++);
                --slots_left;
                ++n_patch;
            } else {
                lane_enter(ctx, n_patch++);
            }
            const int si = std::max(0, i - pad), ei = std::min(H, i + patch + pad);
            const int sj = std::max(0, j - pad), ej = std::min(W, j + patch + pad);
            const int ph = ei - si, pw = ej - sj;
            Slot v;
            set_dense(v, const_cast<float*>(d_in) + (size_t)si * W + sj, 1, 1, ph, pw);
            v.pitch = W;
            v.ps = (long long)H * W;
            v.cs = v.ps;
            float* tmp = (float*)pool_alloc(ctx, (size_t)ph * pw * sizeof(float));
            if (!tmp) { rc_all = fail(ctx, "out of device memory"); break; }
            const int oi = i - si, oj = j - sj;
            const int ch = std::min(patch, std::min(H - i, ph - oi)), cw = std::min(patch, std::min(W - j, pw - oj));
            Rect keep;                     // the pixels of this patch that reach the output image (denoise.py:321)
            keep.y0 = oi; keep.x0 = oj; keep.y1 = oi + ch; keep.x1 = oj + cw; keep.on = true;
            int rc = denoise_region(m, v, tmp, 1, nullptr, split, &keep);
            if (rc == 0) {
                const float* src_ = tmp + (size_t)oi * pw + oj;
                float* dst_ = d_out + (size_t)i * W + j;
                hipError_t e = enqueue(ctx, [=](hipStream_t st) { return launch_copy_box(src_, 0, pw, dst_, 0, W, 1, ch, cw, st); });
                if (e != hipSuccess) rc = fail(ctx, "copy_box failed: %s", hipGetErrorString(e));
            }
            pool_release(ctx, tmp);
            if (rc) rc_all = rc;
        }
    const std::string err = ctx->err;
    if (batched) {
        if (ctx->rec_on && !rc_all && rec_flush(ctx)) rc_all = 1;
        rec_abort(ctx);
    }
    if (lanes_end(ctx) && !rc_all) rc_all = 1;
    if (rc_all && !err.empty()) ctx->err = err;
    return rc_all;
}

int tpz_denoise_2d(tpz_model* m, const float* d_in, int H, int W, int patch, int pad, float* d_out) {
    if (!m || !d_in || !d_out) return fail(m ? m->ctx : nullptr, "tpz_denoise_2d: NULL argument");
    tpz_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int Do, Ho, Wo;
    tpz_model_out_shape(m, 1, 8 * 64, 8 * 64, &Do, &Ho, &Wo);
    if (Ho != 8 * 64 || Wo != 8 * 64) return fail(ctx, "tpz_denoise_2d: the model does not preserve the image size");
    if (m->split_ok && !ctx->exact) {
        // 2xf16 path for the whole micrograph; any activation beyond the f16 range re-runs it on the fp32 kernels
        HIPCHK(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(unsigned), ctx->stream));
        if (denoise_2d_pass(m, d_in, H, W, patch, pad, d_out, true)) return 1;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_flag, ctx->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (*ctx->h_flag == 0) { ++m->n_split; return 0; }
        ++m->n_fallback;
    }
    return denoise_2d_pass(m, d_in, H, W, patch, pad, d_out, false);
}

static int denoise_3d_pass(tpz_model* m, const float* d_in, int D, int H, int W, int patch, int pad, float* d_out,
                           bool split, int shard = 0, int n_shards = 1) {
    tpz_ctx* ctx = m->ctx;
    if (patch < 1) {
        Slot v;
        set_dense(v, const_cast<float*>(d_in), 1, D, H, W);
        return denoise_region(m, v, d_out, 1, nullptr, split);
    }
    // global mean / population std (numpy, denoise.py:343)
    float* g = next_nrm(ctx);
    HIPCHK(ctx, launch_meanstd(d_in, D, H, W, (long long)H * W, W, 0, 0, nullptr, ctx->d_part, PART_BLOCKS, g, ctx->stream));
    const int d = patch + 2 * pad;
    const size_t tn = (size_t)d * d * d;
    // the tiles are independent: batches of `batch` tiles on the 2xf16 path (as the patches of denoise_2d_pass), else the lanes
    const int batch = split ? batch_that_fits(ctx, m, d, d, d) : 0;
    const bool batched = batch >= 2;
    if (lanes_begin(ctx)) return 1;
    // an instance = one tile in flight: (lane, image of the batch) when batched, a lane otherwise
    const int lanes_used = ctx->lanes_on ? ctx->lanes_live : 1;
    const int per_lane = batched ? batch : 1;
    const int n_inst = lanes_used * per_lane;
    float *tiles[N_LANES * SPLIT_MULTI_MAX] = {}, *touts[N_LANES * SPLIT_MULTI_MAX] = {};
    int rc = 0;
    auto inst_enter = [&](int inst) {
        if (batched) {
            ctx->rec_cur = inst % per_lane;
            ctx->pool_cur = &ctx->rec_pools[inst / per_lane][inst % per_lane];
        } else lane_enter(ctx, inst);
    };
    for (int l = 0; l < n_inst; ++l) {
        inst_enter(l);
        tiles[l] = (float*)pool_alloc(ctx, tn * sizeof(float));
        touts[l] = (float*)pool_alloc(ctx, tn * sizeof(float));
        if (!tiles[l] || !touts[l]) rc = fail(ctx, "out of device memory");
    }
    int n_tile = 0, tile_index = 0, slot = 0, slots_left = 0, n_batches = 0;
    for (int i = 0; i < D && !rc; i += patch)
        for (int j = 0; j < H && !rc; j += patch)
            for (int k = 0; k < W && !rc; k += patch) {
                if (tile_index++ % n_shards != shard) continue;        // another rank's tile
                int l = n_tile++ % n_inst;
                if (batched) {
                    if (slots_left == 0) {
                        if (ctx->rec_on && rec_flush(ctx)) { rc = 1; break; }
                        rec_begin(ctx, n_batches++);
                        slots_left = batch;
                        slot = 0;
                    }
                    l = ctx->rec_lane * per_lane + slot++;
                    --slots_left;
                }
                inst_enter(l);
                float *tile = tiles[l], *tout = touts[l];
                hipError_t e = enqueue(ctx, [=](hipStream_t st) {
                    return launch_extract_tile3d(d_in, D, H, W, i - pad, j - pad, k - pad, d, g, tile, st);
                });
                if (e != hipSuccess) { rc = fail(ctx, "extract_tile3d failed: %s", hipGetErrorString(e)); break; }
                Slot tv;
                set_dense(tv, tile, 1, d, d, d);
                // only the centre of the tile is kept (below): every layer computes the box those voxels depend on (need_regions)
                const int pz = std::min(patch, D - i), py = std::min(patch, H - j), px = std::min(patch, W - k);
                Rect keep;
                keep.z0 = pad; keep.z1 = pad + pz; keep.y0 = pad; keep.y1 = pad + py; keep.x0 = pad; keep.x1 = pad + px;
                keep.on = pad > 0;
                rc = denoise_region(m, tv, tout, 2, g, split, &keep);
                if (rc) break;
                {
                    const float* src_ = tout + ((size_t)pad * d + pad) * d + pad;
                    float* dst_ = d_out + ((size_t)i * H + j) * W + k;
                    e = enqueue(ctx, [=](hipStream_t st) {
                        return launch_copy_box(src_, (long long)d * d, d, dst_, (long long)H * W, W, pz, py, px, st);
                    });
                }
                if (e != hipSuccess) rc = fail(ctx, "copy_box failed: %s", hipGetErrorString(e));
            }
    const std::string err = ctx->err;
    if (batched && ctx->rec_on && !rc && rec_flush(ctx)) rc = 1;
    for (int l = 0; l < n_inst; ++l) {
        inst_enter(l);
        if (tiles[l]) pool_release(ctx, tiles[l]);
        if (touts[l]) pool_release(ctx, touts[l]);
    }
    if (batched) rec_abort(ctx);
    if (lanes_end(ctx) && !rc) rc = 1;
    if (rc && !err.empty()) ctx->err = err;
    return rc;
}

int tpz_denoise_3d_shard(tpz_model* m, const float* d_in, int D, int H, int W, int patch, int pad, int shard, int n_shards,
                         float* d_out) {
    if (!m || !d_in || !d_out) return fail(m ? m->ctx : nullptr, "tpz_denoise_3d: NULL argument");
    if (n_shards < 1 || shard < 0 || shard >= n_shards) return fail(m->ctx, "tpz_denoise_3d_shard: shard %d of %d", shard, n_shards);
    if (patch < 1 && n_shards > 1) return fail(m->ctx, "tpz_denoise_3d_shard: an untiled volume cannot be sharded");
    tpz_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (m->split_ok && !ctx->exact) {
        // 2xf16 path for the whole tomogram; any activation beyond the f16 range re-runs it on the fp32 kernels
        HIPCHK(ctx, hipMemsetAsync(ctx->d_flag, 0, sizeof(unsigned), ctx->stream));
        if (denoise_3d_pass(m, d_in, D, H, W, patch, pad, d_out, true, shard, n_shards)) return 1;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_flag, ctx->d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (*ctx->h_flag == 0) { ++m->n_split; return 0; }
        ++m->n_fallback;
    }
    return denoise_3d_pass(m, d_in, D, H, W, patch, pad, d_out, false, shard, n_shards);
}
int tpz_denoise_3d(tpz_model* m, const float* d_in, int D, int H, int W, int patch, int pad, float* d_out) {
    return tpz_denoise_3d_shard(m, d_in, D, H, W, patch, pad, 0, 1, d_out);
}

int tpz_mean_std(tpz_ctx* ctx, const float* d_x, size_t n, int unbiased, float* h_mean_std) {
    if (!ctx || !d_x || !h_mean_std || n == 0) return fail(ctx, "tpz_mean_std: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float* out = next_nrm(ctx);
    // present the vector as rows of <= 2^20 elements so int geometry cannot overflow
    const int Wv = (int)std::min<size_t>(n, (size_t)1 << 20);
    const size_t rows = n / Wv;
    if (rows * (size_t)Wv != n) {
        // fall back to a single row when n is not a multiple (n < 2^31 required)
        if (n >= ((size_t)1 << 31)) return fail(ctx, "tpz_mean_std: n too large for a ragged vector");
        HIPCHK(ctx, launch_meanstd(d_x, 1, 1, (int)n, 0, (int)n, unbiased, 0, nullptr, ctx->d_part, PART_BLOCKS, out, ctx->stream));
    } else {
        HIPCHK(ctx, launch_meanstd(d_x, 1, (int)rows, Wv, 0, Wv, unbiased, 0, nullptr, ctx->d_part, PART_BLOCKS, out, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(h_mean_std, out, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- 2-component Gaussian mixture fit (topaz normalize) ---------------------------------------------------
// topaz/stats.py:87-117 norm_fit + :120-203 gmm_fit (share_var = True), evaluated in fp64 from the sufficient
// statistics of gmm_pass_kernel: one device pass per EM iteration, the scalar M-step on the host.
static double beta_logpdf(double x, double a, double b) {
    // scipy.stats.beta.logpdf: xlog1py(b-1, -x) + xlogy(a-1, x) - betaln(a, b)   (0 * log(0) = 0)
    const double t1 = (b - 1.0) == 0.0 ? 0.0 : (b - 1.0) * std::log1p(-x);
    const double t2 = (a - 1.0) == 0.0 ? 0.0 : (a - 1.0) * std::log(x);
    return t1 + t2 - (std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b));
}

int tpz_gmm_fit(tpz_ctx* ctx, const float* d_x, size_t n, const double* pis, const double* splits, int n_init,
                double alpha, double beta, double scale, int num_iters, double tol, double* mus, double* stds,
                double* pis_out, double* logps) {
    if (!ctx || !d_x || !pis || !splits || !mus || !stds || !pis_out || !logps || n < 2 || n_init < 1)
        return fail(ctx, "tpz_gmm_fit: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *d_par = nullptr, *d_out = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&d_par, 8 * sizeof(double)));
    if (hipMalloc((void**)&d_out, 8 * sizeof(double)) != hipSuccess) { (void)hipFree(d_par); return fail(ctx, "hipMalloc failed"); }
    int rc = 0;
    const double N = (double)n;
    auto pass = [&](int mode, const double (&par)[6], double (&S)[7]) -> int {
        if (hipMemcpyAsync(d_par, par, 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return 1;
        prof_begin(ctx, 2, 0);
        hipError_t e = launch_gmm_pass(d_x, n, mode, d_par, ctx->d_part, 256, d_out, ctx->stream);
        prof_end(ctx);
        if (e != hipSuccess) return 1;
        if (hipMemcpyAsync(S, d_out, 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return 1;
        return hipStreamSynchronize(ctx->stream) != hipSuccess;
    };
    // global moments: hard pass with split = +inf puts everything in component 0
    double S[7];
    {
        const double par[6] = {INFINITY, 0, 0, 0, 0, 0};
        if (pass(0, par, S)) rc = fail(ctx, "tpz_gmm_fit: device pass failed");
    }
    const double mu_all = S[3] / N;
    const double var_unbiased = (S[5] - 2.0 * mu_all * S[3] + mu_all * mu_all * N) / (N - 1.0);   // torch .var()
    for (int i = 0; i < n_init && !rc; ++i) {
        double pi = pis[i];
        if (pi == 1.0) {
            // single-component model (stats.py:100-103): note the reference adds beta.PDF(1), not its logarithm
            const double pdf1 = beta == 1.0 ? std::exp(-(std::lgamma(alpha) + std::lgamma(beta) - std::lgamma(alpha + beta)))
                                            : (beta > 1.0 ? 0.0 : INFINITY);
            logps[i] = scale * (-(N - 1.0) / 2.0 - N * 0.5 * std::log(2.0 * M_PI * var_unbiased)) + pdf1;
            mus[i] = mu_all;
            stds[i] = std::sqrt(var_unbiased);
            pis_out[i] = 1.0;
            continue;
        }
        auto m_step = [&](const double (&T)[7], double& mu0, double& mu1, double& var) {
            mu0 = T[1] > 0 ? T[3] / T[1] : mu_all;
            mu1 = T[2] > 0 ? T[4] / T[2] : mu_all;
            var = ((T[5] - 2.0 * mu0 * T[3] + mu0 * mu0 * T[1]) + (T[6] - 2.0 * mu1 * T[4] + mu1 * mu1 * T[2])) / N;
        };
        double mu0, mu1, var;
        {
            const double par[6] = {splits[i], 0, 0, 0, 0, 0};
            if (pass(0, par, S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
        }
        m_step(S, mu0, mu1, var);
        auto e_step = [&](double (&T)[7]) -> int {
            const double par[6] = {mu0, mu1, var, var, std::log1p(-pi), std::log(pi)};
            return pass(1, par, T);
        };
        if (e_step(S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
        double logp = scale * S[0] + beta_logpdf(pi, alpha, beta);
        double logp_cur = logp;
        for (int it = 1; it <= num_iters; ++it) {
            // M-step from the assignments of the last E-step (S), MAP estimate of pi under the Beta prior
            const double a_ = alpha + S[2], b_ = beta + N - S[2];
            pi = (a_ - 1.0) / (a_ + b_ - 2.0);
            m_step(S, mu0, mu1, var);
            if (e_step(S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
            logp = scale * S[0] + beta_logpdf(pi, alpha, beta);
            if (logp - logp_cur <= tol) break;
            logp_cur = logp;
        }
        logps[i] = logp;
        mus[i] = mu1;
        stds[i] = std::sqrt(var);
        pis_out[i] = pi;
    }
    (void)hipFree(d_par);
    (void)hipFree(d_out);
    return rc;
}

int tpz_affine(tpz_ctx* ctx, const float* d_x, size_t n, float scale, float shift, float* d_y) {
    if (!ctx || !d_x || !d_y) return fail(ctx, "tpz_affine: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prof_begin(ctx, 2, 0);
    hipError_t e = launch_affine(d_x, d_y, n, scale, shift, ctx->stream);
    prof_end(ctx);
    HIPCHK(ctx, e);
    return 0;
}

int tpz_normalize(tpz_ctx* ctx, const float* d_x, size_t n, float mean, float std, float* d_y) {
    if (!ctx || !d_x || !d_y) return fail(ctx, "tpz_normalize: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prof_begin(ctx, 2, 0);
    hipError_t e = launch_normalize(d_x, d_y, n, mean, std, ctx->stream);
    prof_end(ctx);
    HIPCHK(ctx, e);
    return 0;
}

// ---- single ops --------------------------------------------------------------------------------
int tpz_conv(tpz_ctx* ctx, int dims, const float* d_in, int cin1, int D1, int H1, int W1, const float* d_in2, int cin,
             int D, int H, int W, const float* h_w, const float* h_b, int cout, int k, int dil, int pad, float slope,
             const float* d_res, int res_crop, const float* h_post_scale, const float* h_post_shift,
             const float* h_head_w, float head_b, float* d_out) {
    if (!ctx || !d_in || !h_w || !d_out) return fail(ctx, "tpz_conv: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t taps = dims == 3 ? (size_t)k * k * k : (size_t)k * k;
    std::vector<float> blob;
    tpz_layer L;
    memset(&L, 0, sizeof L);
    L.op = TPZ_OP_CONV; L.dims = dims; L.src = 0; L.src2 = d_in2 ? 1 : -1; L.dst = 3;
    L.cin = cin; L.cout = cout; L.k = k; L.dil = dil; L.pad = pad; L.slope = slope;
    L.res = d_res ? 2 : -1; L.res_crop = res_crop;
    L.w_off = 0;
    blob.insert(blob.end(), h_w, h_w + (size_t)cout * cin * taps);
    L.b_off = -1;
    if (h_b) { L.b_off = (int64_t)blob.size(); blob.insert(blob.end(), h_b, h_b + cout); }
    L.post_scale_off = L.post_shift_off = -1;
    if (h_post_scale && h_post_shift) {
        L.post_scale_off = (int64_t)blob.size(); blob.insert(blob.end(), h_post_scale, h_post_scale + cout);
        L.post_shift_off = (int64_t)blob.size(); blob.insert(blob.end(), h_post_shift, h_post_shift + cout);
    }
    L.head = h_head_w ? 1 : 0;
    if (h_head_w) {
        L.head_w_off = (int64_t)blob.size(); blob.insert(blob.end(), h_head_w, h_head_w + cout);
        L.head_b_off = (int64_t)blob.size(); blob.push_back(head_b);
    }
    tpz_model* m = nullptr;
    if (model_load(ctx, &L, 1, blob.data(), blob.size(), {cin1, d_in2 ? cin - cin1 : 0, cout}, &m)) return 1;
    std::vector<Slot> slots(4);
    set_dense(slots[0], const_cast<float*>(d_in), cin1, D1, H1, W1);
    if (d_in2) set_dense(slots[1], const_cast<float*>(d_in2), cin - cin1, D, H, W);
    if (d_res) {
        const int span = dil * (k - 1);
        const int Ho = H + 2 * pad - span, Wo = W + 2 * pad - span, Do = dims == 3 ? D + 2 * pad - span : 1;
        set_dense(slots[2], const_cast<float*>(d_res), cout, dims == 3 ? Do + 2 * res_crop : 1, Ho + 2 * res_crop,
                  Wo + 2 * res_crop);
    }
    m->n_slots = 4;
    m->last_use.assign(4, 0);
    int rc = run_program(m, slots, d_out, nullptr);
    if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, "tpz_conv: kernel failed");
    tpz_model_free(m);
    return rc;
}

int tpz_maxpool2(tpz_ctx* ctx, int dims, const float* d_in, int C, int D, int H, int W, float* d_out) {
    if (!ctx || !d_in || !d_out) return fail(ctx, "tpz_maxpool2: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_maxpool2(d_in, d_out, C, D, H, W, dims, ctx->stream));
    return 0;
}

int tpz_transpose_2d(tpz_ctx* ctx, const float* d_in, int rows, int cols, float* d_out) {
    if (!ctx || !d_in || !d_out) return fail(ctx, "tpz_transpose_2d: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prof_begin(ctx, 2, 0);
    hipError_t e = launch_transpose(d_in, d_out, rows, cols, ctx->stream);
    prof_end(ctx);
    HIPCHK(ctx, e);
    return 0;
}

int tpz_filter_2d(tpz_ctx* ctx, const float* d_in, int H, int W, const float* h_w, int k, float bias, float* d_out) {
    if (!ctx || !d_in || !h_w || !d_out || k < 1 || (k & 1) == 0) return fail(ctx, "tpz_filter_2d: bad arguments");
    return tpz_conv(ctx, 2, d_in, 1, 1, H, W, nullptr, 1, 1, H, W, h_w, &bias, 1, k, 1, k / 2, 1.0f, nullptr, 0, nullptr,
                    nullptr, nullptr, 0.f, d_out);
}

// ---- staging ring: pinned host buffer + device buffer + events per slot, one copy stream ----------------------
struct tpz_stage {
    tpz_ctx* ctx = nullptr;
    size_t slot_bytes = 0;
    hipStream_t copy = nullptr;
    struct Slot {
        void* h = nullptr;
        void* d = nullptr;
        hipEvent_t ready = nullptr;       // last copy of this slot (either direction) finished
        hipEvent_t released = nullptr;    // kernels reading / writing the device buffer are done with it
        hipEvent_t produced = nullptr;    // the result to copy back exists
    };
    std::vector<Slot> slots;
};

int tpz_stage_create(tpz_ctx* ctx, size_t slot_bytes, int depth, tpz_stage** out) {
    if (!ctx || !out || slot_bytes == 0 || depth < 1 || depth > 64) return fail(ctx, "tpz_stage_create: bad arguments");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    tpz_stage* st = new tpz_stage();
    st->ctx = ctx;
    st->slot_bytes = slot_bytes;
    st->slots.resize(depth);
    bool ok = hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking) == hipSuccess;
    for (auto& sl : st->slots) {
        ok = ok && hipHostMalloc(&sl.h, slot_bytes) == hipSuccess && hipMalloc(&sl.d, slot_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.released, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.produced, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { tpz_stage_free(st); return fail(ctx, "tpz_stage_create: allocation of %d x %zu bytes failed", depth, slot_bytes); }
    *out = st;
    return 0;
}
void tpz_stage_free(tpz_stage* st) {
    if (!st) return;
    (void)hipSetDevice(st->ctx->device);
    if (st->copy) (void)hipStreamSynchronize(st->copy);
    (void)hipStreamSynchronize(st->ctx->stream);
    for (auto& sl : st->slots) {
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d) (void)hipFree(sl.d);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.released) (void)hipEventDestroy(sl.released);
        if (sl.produced) (void)hipEventDestroy(sl.produced);
    }
    if (st->copy) (void)hipStreamDestroy(st->copy);
    delete st;
}
static tpz_stage::Slot* stage_slot(tpz_stage* st, int slot) {
    return (st && slot >= 0 && slot < (int)st->slots.size()) ? &st->slots[slot] : nullptr;
}
void* tpz_stage_host_ptr(tpz_stage* st, int slot) { auto* s = stage_slot(st, slot); return s ? s->h : nullptr; }
void* tpz_stage_device_ptr(tpz_stage* st, int slot) { auto* s = stage_slot(st, slot); return s ? s->d : nullptr; }
int tpz_stage_h2d(tpz_stage* st, int slot, const void* h_src, size_t bytes) {
    auto* sl = stage_slot(st, slot);
    if (!sl || bytes > st->slot_bytes) return fail(st ? st->ctx : nullptr, "tpz_stage_h2d: bad slot or size");
    tpz_ctx* ctx = st->ctx;
    if (h_src && h_src != sl->h) {
        // the pinned buffer may still be the source / target of this slot's previous copy
        HIPCHK(ctx, hipEventSynchronize(sl->ready));
        memcpy(sl->h, h_src, bytes);
    }
    HIPCHK(ctx, hipStreamWaitEvent(st->copy, sl->released, 0));      // kernels of the slot's previous use are done
    HIPCHK(ctx, hipMemcpyAsync(sl->d, sl->h, bytes, hipMemcpyHostToDevice, st->copy));
    HIPCHK(ctx, hipEventRecord(sl->ready, st->copy));
    return 0;
}
int tpz_stage_acquire(tpz_stage* st, int slot) {
    auto* sl = stage_slot(st, slot);
    if (!sl) return fail(st ? st->ctx : nullptr, "tpz_stage_acquire: bad slot");
    HIPCHK(st->ctx, hipStreamWaitEvent(st->ctx->stream, sl->ready, 0));
    return 0;
}
int tpz_stage_release(tpz_stage* st, int slot) {
    auto* sl = stage_slot(st, slot);
    if (!sl) return fail(st ? st->ctx : nullptr, "tpz_stage_release: bad slot");
    HIPCHK(st->ctx, hipEventRecord(sl->released, st->ctx->stream));
    return 0;
}
int tpz_stage_d2h(tpz_stage* st, int slot, const void* d_src, size_t bytes) {
    auto* sl = stage_slot(st, slot);
    if (!sl || !d_src || bytes > st->slot_bytes) return fail(st ? st->ctx : nullptr, "tpz_stage_d2h: bad slot or size");
    tpz_ctx* ctx = st->ctx;
    HIPCHK(ctx, hipEventRecord(sl->produced, ctx->stream));           // everything queued so far produced d_src
    HIPCHK(ctx, hipStreamWaitEvent(st->copy, sl->produced, 0));
    HIPCHK(ctx, hipMemcpyAsync(sl->h, d_src, bytes, hipMemcpyDeviceToHost, st->copy));
    HIPCHK(ctx, hipEventRecord(sl->ready, st->copy));
    return 0;
}
int tpz_stage_wait(tpz_stage* st, int slot) {
    auto* sl = stage_slot(st, slot);
    if (!sl) return fail(st ? st->ctx : nullptr, "tpz_stage_wait: bad slot");
    HIPCHK(st->ctx, hipEventSynchronize(sl->ready));
    return 0;
}

// host-pointer entry points: two slots of the ctx's own ring (input, output), grown on demand
static int io_stage(tpz_ctx* ctx, size_t bytes, tpz_stage** out) {
    if (ctx->io_stage && ctx->io_stage->slot_bytes < bytes) { tpz_stage_free(ctx->io_stage); ctx->io_stage = nullptr; }
    if (!ctx->io_stage && tpz_stage_create(ctx, (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1), 2, &ctx->io_stage)) return 1;
    *out = ctx->io_stage;
    return 0;
}
int tpz_score_2d_host(tpz_model* m, const float* h_in, int H, int W, float* h_out_logits) {
    if (!m || !h_in || !h_out_logits || H < 1 || W < 1) return fail(m ? m->ctx : nullptr, "tpz_score_2d_host: bad arguments");
    tpz_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int Do, Ho, Wo;
    tpz_model_out_shape(m, 1, H, W, &Do, &Ho, &Wo);
    if (Ho < 1 || Wo < 1) return fail(ctx, "input %dx%d too small for this model", H, W);
    const size_t nin = (size_t)H * W * sizeof(float), nout = (size_t)Ho * Wo * sizeof(float);
    tpz_stage* st;
    if (io_stage(ctx, std::max(nin, nout), &st)) return 1;
    if (tpz_stage_h2d(st, 0, h_in, nin) || tpz_stage_acquire(st, 0)) return 1;
    if (tpz_model_forward(m, (const float*)tpz_stage_device_ptr(st, 0), 1, 1, H, W, (float*)tpz_stage_device_ptr(st, 1))) return 1;
    if (tpz_stage_release(st, 0) || tpz_stage_d2h(st, 1, tpz_stage_device_ptr(st, 1), nout) || tpz_stage_wait(st, 1)) return 1;
    memcpy(h_out_logits, tpz_stage_host_ptr(st, 1), nout);
    return tpz_stage_release(st, 1);
}
int tpz_denoise_2d_host(tpz_model* m, const float* h_in, int H, int W, int patch, int pad, float* h_out) {
    if (!m || !h_in || !h_out || H < 1 || W < 1) return fail(m ? m->ctx : nullptr, "tpz_denoise_2d_host: bad arguments");
    tpz_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nb = (size_t)H * W * sizeof(float);
    tpz_stage* st;
    if (io_stage(ctx, nb, &st)) return 1;
    if (tpz_stage_h2d(st, 0, h_in, nb) || tpz_stage_acquire(st, 0)) return 1;
    if (tpz_denoise_2d(m, (const float*)tpz_stage_device_ptr(st, 0), H, W, patch, pad, (float*)tpz_stage_device_ptr(st, 1))) return 1;
    if (tpz_stage_release(st, 0) || tpz_stage_d2h(st, 1, tpz_stage_device_ptr(st, 1), nb) || tpz_stage_wait(st, 1)) return 1;
    memcpy(h_out, tpz_stage_host_ptr(st, 1), nb);
    return tpz_stage_release(st, 1);
}
int tpz_nms_2d_host(tpz_ctx* ctx, const float* h_score, int H, int W, int r, float threshold, int32_t* h_coords,
                    float* h_scores, int cap, int* h_n) {
    if (!ctx || !h_score || !h_coords || !h_scores || H < 1 || W < 1 || cap < 0) return fail(ctx, "tpz_nms_2d_host: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nb = (size_t)H * W * sizeof(float), ob = (size_t)std::max(cap, 1) * 3 * sizeof(float);
    tpz_stage* st;
    if (io_stage(ctx, std::max(nb, ob), &st)) return 1;
    if (tpz_stage_h2d(st, 0, h_score, nb) || tpz_stage_acquire(st, 0)) return 1;
    int32_t* d_coords = (int32_t*)tpz_stage_device_ptr(st, 1);
    float* d_scores = (float*)tpz_stage_device_ptr(st, 1) + (size_t)2 * std::max(cap, 1);
    int n = 0;
    const int rc = tpz_nms_2d(ctx, (const float*)tpz_stage_device_ptr(st, 0), H, W, r, threshold, d_coords, d_scores, cap, &n);
    if (h_n) *h_n = n;
    if (tpz_stage_release(st, 0)) return 1;
    if (rc) return rc;
    if (tpz_stage_d2h(st, 1, tpz_stage_device_ptr(st, 1), ob) || tpz_stage_wait(st, 1)) return 1;
    const size_t k = (size_t)std::min(n, cap);
    memcpy(h_coords, tpz_stage_host_ptr(st, 1), k * 2 * sizeof(int32_t));
    memcpy(h_scores, (const float*)tpz_stage_host_ptr(st, 1) + (size_t)2 * std::max(cap, 1), k * sizeof(float));
    return tpz_stage_release(st, 1);
}

// ---- NMS ------------------------------------------------------------------------------------------
// Device-side counters of one NMS call: [0 .. NMS_BATCH] lengths of the candidate lists (sweep k of a batch reads [k] and
// appends its leftovers under [k + 1]); [NMS_VER + k] length of sweep k's verify list; [NMS_SNAP + k] picks before sweep k
// of the batch (the push after sweep k covers keys[snap[k] .. snap[k + 1])); [NMS_PICKS] picks so far.
// h_aux: n_aux "near" entries (phase A of a sweep) followed by n_aux2 entries of the whole suppression set (phase B, push):
// 2-D cells dy * 65536 + (dx + 32768), 3-D flat-index deltas.
static int nms_common(tpz_ctx* ctx, const float* d_score, int D, int H, int W, int dims, int r, const int* h_aux,
                      int n_aux, int n_aux2, float threshold, int32_t* d_coords, float* d_scores, int cap, int* h_n) {
    const size_t n = (size_t)D * H * W;
    if (n >= ((size_t)1 << 32)) return fail(ctx, "nms: more than 2^32 elements");
    hipStream_t s = ctx->stream;
    // capacity of the pick list: every pick suppresses at least itself and (r >= 1) its row neighbours, but the bound that
    // always holds is one pick per pixel; keys are only ever touched up to the pick count
    size_t kcap = 4096;
    while (kcap < n) kcap <<= 1;
    uint8_t* status = (uint8_t*)pool_alloc(ctx, n);
    uint32_t* listA = (uint32_t*)pool_alloc(ctx, n * sizeof(uint32_t));
    uint32_t* listB = (uint32_t*)pool_alloc(ctx, n * sizeof(uint32_t));
    uint32_t* listV = (uint32_t*)pool_alloc(ctx, n * sizeof(uint32_t));
    uint64_t* keys = (uint64_t*)pool_alloc(ctx, kcap * sizeof(uint64_t));
    int* d_aux = (int*)pool_alloc(ctx, std::max(1, n_aux + n_aux2) * sizeof(int));
    auto done = [&](int code) {
        pool_release(ctx, status);
        pool_release(ctx, listA);
        pool_release(ctx, listB);
        pool_release(ctx, listV);
        pool_release(ctx, keys);
        pool_release(ctx, d_aux);
        return code;
    };
    if (!status || !listA || !listB || !listV || !keys || !d_aux) return done(fail(ctx, "nms: out of device memory"));
    int rc = 0;
    unsigned int* cnt = ctx->d_counters;
    prof_begin(ctx, 3, 0);
    auto bail = [&](const char* what, hipError_t e) {
        prof_end(ctx);
        return done(fail(ctx, "nms: %s failed: %s", what, hipGetErrorString(e)));
    };
    hipError_t e = hipMemsetAsync(cnt, 0, NMS_COUNTERS * sizeof(unsigned int), s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_aux, h_aux, (n_aux + n_aux2) * sizeof(int), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = nms_mark(d_score, n, threshold, status, listA, cnt, s);       // candidates -> listA, count -> cnt[0]
    if (e != hipSuccess) return bail("mark phase", e);
    // Relaxation sweeps until no candidate is undecided.  Every sweep decides at least the highest-priority undecided
    // candidate, so the candidate count bounds the sweep count; in practice a map needs 5 - 10.  The sweeps of a batch are
    // queued back to back (their list lengths stay on the device) and one counter is read per batch.
    unsigned int hc[NMS_COUNTERS];
    unsigned long long sweeps = 0;
    unsigned int ncand = 0;
    bool first_batch = true;
    size_t hint = n;                         // upper bound of the current list's length (grid sizing only)
    uint32_t *lin = listA, *lout = listB;
    unsigned int npicks = 0;
    for (;;) {
        int k = 0;
        for (; k < NMS_BATCH; ++k) {
            e = dims == 2 ? nms2d_sweep(d_score, H, W, d_aux, n_aux, d_aux + n_aux, n_aux2, status, lin, lout, listV, cnt + k,
                                        cnt + NMS_VER + k, cnt + NMS_SNAP + k, keys, cnt + NMS_PICKS, hint, s)
                          : nms3d_sweep(d_score, (long long)n, d_aux, n_aux, d_aux + n_aux, n_aux2, status, lin, lout, listV, cnt + k,
                                        cnt + NMS_VER + k, cnt + NMS_SNAP + k, keys, cnt + NMS_PICKS, hint, s);
            if (e != hipSuccess) return bail("sweep", e);
            std::swap(lin, lout);
        }
        sweeps += NMS_BATCH;
        e = hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("sweep batch", e);
        npicks = hc[NMS_PICKS];
        const unsigned int remaining = hc[NMS_BATCH];
        if (remaining == 0) break;
        if (first_batch) { ncand = hc[0]; first_batch = false; }
        if (sweeps > 2ull * ncand + NMS_BATCH) { prof_end(ctx); return done(fail(ctx, "nms: fix-point did not converge")); }
        // next batch: the leftovers are list [NMS_BATCH] -> restart the chain at [0] with that length
        hint = remaining;
        e = hipMemcpyAsync(cnt, cnt + NMS_BATCH, sizeof(unsigned int), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(cnt + 1, 0, 2 * NMS_BATCH * sizeof(unsigned int), s);      // chain + verify counts
        if (e == hipSuccess) e = hipMemcpyAsync(cnt + NMS_SNAP, cnt + NMS_SNAP + NMS_BATCH, sizeof(unsigned int), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return bail("sweep chain reset", e);
    }
    if (npicks > 0) {
        if ((size_t)npicks > kcap) { prof_end(ctx); return done(fail(ctx, "nms: pick list overflow")); }
        size_t sp2 = 4096;
        while (sp2 < npicks) sp2 <<= 1;
        e = fill_u64(keys, npicks, sp2, 0ull, s);
        if (e == hipSuccess) e = bitonic_sort_desc(keys, sp2, s);
        const unsigned int nw = std::min<unsigned int>(npicks, (unsigned int)std::max(cap, 0));
        if (e == hipSuccess) e = nms_write(keys, nw, d_score, H, W, dims, d_coords, d_scores, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("sort/write", e);
    }
    prof_end(ctx);
    if (h_n) *h_n = (int)npicks;
    if ((long long)npicks > (long long)cap) rc = fail(ctx, "nms: %u picks exceed the output capacity %d", npicks, cap);
    return done(rc);
}

int tpz_nms_2d(tpz_ctx* ctx, const float* d_score, int H, int W, int r, float threshold, int32_t* d_coords,
               float* d_scores, int cap, int* h_n) {
    if (!ctx || !d_score || H < 1 || W < 1 || r < 0) return fail(ctx, "tpz_nms_2d: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (r > 16000) return fail(ctx, "tpz_nms_2d: radius too large");
    // the suppression disk ii^2 + jj^2 <= r^2 as (dy, dx) cells, dy * 65536 + (dx + 32768); the "near" subset (phase A of a
    // sweep) is its intersection with the 5 x 5 neighbourhood, nearest first, without the centre
    std::vector<std::pair<int, int>> near;
    std::vector<int> aux, full;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            if (dy * dy + dx * dx > r * r) continue;
            full.push_back(dy * 65536 + (dx + 32768));
            if ((dy || dx) && std::abs(dy) <= 2 && std::abs(dx) <= 2) near.push_back({dy * dy + dx * dx, dy * 65536 + (dx + 32768)});
        }
    std::sort(near.begin(), near.end());
    for (auto& c : near) aux.push_back(c.second);
    const int n_near = (int)aux.size();
    aux.insert(aux.end(), full.begin(), full.end());
    return nms_common(ctx, d_score, 1, H, W, 2, r, aux.data(), n_near, (int)full.size(), threshold, d_coords, d_scores, cap, h_n);
}

int tpz_nms_3d(tpz_ctx* ctx, const float* d_score, int D, int H, int W, int r, double scale, float threshold,
               int32_t* d_coords, float* d_scores, int cap, int* h_n) {
    if (!ctx || !d_score || D < 1 || H < 1 || W < 1 || r < 0) return fail(ctx, "tpz_nms_3d: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // algorithms.py:68-79: r = scale*r (float), width = ceil(r), deltas over the ball
    const double rr = scale * (double)r;
    const int width = (int)std::ceil(rr);
    const long long zs = (long long)H * W, ys = W;
    std::vector<int> deltas;
    std::vector<std::pair<int, int>> near;
    for (int ii = -width; ii <= width; ++ii)
        for (int jj = -width; jj <= width; ++jj)
            for (int kk = -width; kk <= width; ++kk)
                if ((double)(ii * ii + jj * jj + kk * kk) <= rr * rr) {
                    const long long dlt = ii * zs + jj * ys + kk;
                    if (std::llabs(dlt) >= ((long long)1 << 31)) continue;
                    deltas.push_back((int)dlt);
                    if (dlt != 0 && std::abs(ii) <= 1 && std::abs(jj) <= 1 && std::abs(kk) <= 1)
                        near.push_back({ii * ii + jj * jj + kk * kk, (int)dlt});
                }
    std::sort(deltas.begin(), deltas.end());
    deltas.erase(std::unique(deltas.begin(), deltas.end()), deltas.end());
    std::sort(near.begin(), near.end());
    std::vector<int> aux;
    for (auto& c : near)
        if (std::find(aux.begin(), aux.end(), c.second) == aux.end()) aux.push_back(c.second);
    const int n_near = (int)aux.size();
    aux.insert(aux.end(), deltas.begin(), deltas.end());
    return nms_common(ctx, d_score, D, H, W, 3, r, aux.data(), n_near, (int)deltas.size(), threshold, d_coords, d_scores, cap, h_n);
}

// ---- profiling -----------------------------------------------------------------------------------
int tpz_prof_enable(tpz_ctx* ctx, int on) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    prof_flush(ctx);
    ctx->prof = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return 0;
}
int tpz_prof_reset(tpz_ctx* ctx) {
    if (!ctx) return fail(nullptr, "ctx is NULL");
    prof_flush(ctx);
    for (int i = 0; i < 4; ++i) { ctx->acc_ms[i] = 0; ctx->acc_n[i] = 0; ctx->acc_flops[i] = 0; }
    ctx->per_kernel.clear();
    return 0;
}
int tpz_prof_get_kernel(tpz_ctx* ctx, int rank, double* ms, long long* launches, double* flops, char* name,
                        int name_len) {
    if (!ctx || rank < 0) return fail(ctx, "tpz_prof_get_kernel: bad arguments");
    prof_flush(ctx);
    std::vector<std::pair<const void*, ProfAcc>> order(ctx->per_kernel);
    std::stable_sort(order.begin(), order.end(),
                     [](const std::pair<const void*, ProfAcc>& x, const std::pair<const void*, ProfAcc>& y) {
                         return x.second.ms > y.second.ms;
                     });
    const bool have = rank < (int)order.size();
    const ProfAcc acc = have ? order[rank].second : ProfAcc();
    if (ms) *ms = acc.ms;
    if (launches) *launches = acc.n;
    if (flops) *flops = acc.flops;
    if (name && name_len > 0) {
        name[0] = 0;
        if (have) snprintf(name, name_len, "%s", (const char*)order[rank].first);
    }
    return 0;
}
int tpz_prof_get_kernel_bytes(tpz_ctx* ctx, int rank, double* bytes) {
    if (!ctx || rank < 0 || !bytes) return fail(ctx, "tpz_prof_get_kernel_bytes: bad arguments");
    prof_flush(ctx);
    std::vector<std::pair<const void*, ProfAcc>> order(ctx->per_kernel);
    std::stable_sort(order.begin(), order.end(),
                     [](const std::pair<const void*, ProfAcc>& x, const std::pair<const void*, ProfAcc>& y) {
                         return x.second.ms > y.second.ms;
                     });
    *bytes = rank < (int)order.size() ? order[rank].second.bytes : 0.0;
    return 0;
}
int tpz_prof_mfma_sustained(tpz_ctx* ctx, int ms, int zero_operands, double* tflops, double* clock_ratio) {
    if (!ctx || ms < 1 || ms > 5000 || !tflops) return fail(ctx, "tpz_prof_mfma_sustained: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int n_wg = 2 * prop.multiProcessorCount;           // two 4-wave workgroups per CU: two waves per SIMD
    const size_t n_src = 4096 * 8;
    std::vector<_Float16> h(n_src);
    unsigned lcg = 12345u;
    for (auto& v : h) {
        lcg = lcg * 1664525u + 1013904223u;
        v = zero_operands ? (_Float16)0.f : (_Float16)(((int)(lcg >> 8) % 2001 - 1000) * 1e-3f);
    }
    void* d_src = nullptr; float* d_out = nullptr; unsigned long long* d_ticks = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto run = [&](int iters, float* t_ms) {
        hipError_t e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) e = launch_mfma_spin(d_src, d_out, n_wg, iters, d_ticks, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(t_ms, e0, e1);
        return e;
    };
    hipError_t e = hipMalloc(&d_src, n_src * sizeof(_Float16));
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, (size_t)n_wg * 256 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&d_ticks, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpy(d_src, h.data(), n_src * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float t = 0.f;
    const int probe = 20000;
    if (e == hipSuccess) e = run(probe, &t);                 // sizes the loop (and is the first of the warm-up)
    if (e == hipSuccess) {
        const int iters = (int)std::min(2.0e9, std::max(1000.0, probe * (double)ms / std::max(t, 1e-3f)));
        e = run(iters, &t);                                  // the power management settles within this one
        if (e == hipSuccess) e = run(iters, &t);
        unsigned long long ticks[2] = {};
        if (e == hipSuccess) e = hipMemcpy(ticks, d_ticks, sizeof(ticks), hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            const double flop = 8.0 * 16 * 16 * 32 * 2 * (double)iters * n_wg * 4;
            *tflops = flop / (t * 1e-3) * 1e-12;
            // s_memtime over s_memrealtime: proportional to the shader clock (both are read inside the loop's bracket)
            if (clock_ratio) *clock_ratio = ticks[1] ? (double)ticks[0] / (double)ticks[1] : 0.0;
        }
    }
    if (e != hipSuccess) rc = fail(ctx, "tpz_prof_mfma_sustained: %s", hipGetErrorString(e));
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_src); (void)hipFree(d_out); (void)hipFree(d_ticks);
    return rc;
}
int tpz_prof_get_dominant(tpz_ctx* ctx, double* ms, long long* launches, double* flops, char* name, int name_len) {
    return tpz_prof_get_kernel(ctx, 0, ms, launches, flops, name, name_len);
}
int tpz_prof_get(tpz_ctx* ctx, int cls, double* ms, long long* launches, double* flops) {
    if (!ctx || cls < 0 || cls > 3) return fail(ctx, "tpz_prof_get: bad arguments");
    prof_flush(ctx);
    if (ms) *ms = ctx->acc_ms[cls];
    if (launches) *launches = ctx->acc_n[cls];
    if (flops) *flops = ctx->acc_flops[cls];
    return 0;
}

}  // extern "C"
