// NMS driver (kernels: nms.hip).
#include "rt_internal.h"

extern "C" {
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

}  // extern "C"
