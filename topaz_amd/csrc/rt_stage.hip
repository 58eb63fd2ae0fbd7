// Staging ring (pinned host buffer + device buffer + events per slot, one copy stream) and the host-pointer entry points.
#include "rt_internal.h"

extern "C" {
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

}  // extern "C"
