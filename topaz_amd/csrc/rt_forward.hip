// Scoring drivers (tpz_model_forward: range-scaled pass, internal tiling) and the single-op entry points.
#include "rt_internal.h"

extern "C" {
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

}  // extern "C"
