// Denoising drivers: Denoise._denoise on a region, the patched 2-D pass and the tiled 3-D pass (batched over the lanes).
#include "rt_internal.h"

extern "C" {
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

}  // extern "C"
