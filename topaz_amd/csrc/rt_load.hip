// Model loading: kernel choice per layer, weight packing, decoder phase forms, folded projections, widened programs.
#include "rt_internal.h"

int upload(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out) {
    float* d = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&d, std::max<size_t>(n, 4) * sizeof(float)));
    HIPCHK(ctx, hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    if (m) m->dev_allocs.push_back(d);
    *out = d;
    return 0;
}

// per-channel vectors (bias, BN affine, head weights, weight scales) are zero-padded to whole 128-channel tiles plus one,
// so the epilogues fetch them as unclamped float4 loads; zero scale / bias make the padded channels come out as 0
int upload_chan(tpz_ctx* ctx, tpz_model* m, const float* h, size_t n, float** out) {
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
        if (!m->dbg.no_phase && prepare_phases(ctx, m, L, w, c1, c2, rt)) return 1;
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
void pack_weights_split(const SplitKernelInfo& ki, const float* w, int cout, int cin, int n_cog, int n_chunks,
                        std::vector<uint16_t>& out, std::vector<float>& wscale_inv, const float* wp, int cin_b, const float* mul) {
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

const SplitKernelInfo* pick_split(int k, int dil, int cout, int epi, int kx) {
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
        const bool no_srcmajor = m->dbg.no_srcmajor;
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
            const bool no_valu_last = m->dbg.no_valu_last;
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
            const bool no_rw = m->dbg.no_rw;
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
        const bool no_pool3d = m->dbg.no_pool3d;
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
    const bool no_fold = m->dbg.no_fold;
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

// (the tpz_* entry points below take their C linkage from their declarations in include/topaz_hip.h)
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
    // the slots whose channels ARE the program's output channels: the last layer's destination and, through trailing pools
    // (a partial program that ends in MAXPOOL / MAXPOOL2), the tensors it is pooled from -- their producer keeps its width
    std::vector<char> is_out(max_slot + 1, 0);
    is_out[layers[n_layers - 1].dst] = 1;
    for (int i = n_layers - 1; i >= 0 && layers[i].op != TPZ_OP_CONV; --i)
        if (layers[i].dst >= 0 && layers[i].dst <= max_slot && is_out[layers[i].dst] && layers[i].src >= 0 && layers[i].src <= max_slot)
            is_out[layers[i].src] = 1;
    for (int i = 0; i < n_layers; ++i) {
        tpz_layer& L = out_l[i];
        if (L.src < 0 || L.src > max_slot || L.dst <= 0) return false;              // (model_load reports the bad program)
        const int c1 = chan[L.src], c2 = L.src2 >= 0 ? chan[L.src2] : 0;
        const int p1 = pch[L.src], p2 = L.src2 >= 0 ? pch[L.src2] : 0;
        if (L.op != TPZ_OP_CONV) { chan[L.dst] = c1; pch[L.dst] = p1; continue; }
        if (L.cin != c1 + c2 || L.w_off < 0) return false;
        const size_t taps = L.dims == 3 ? (size_t)L.k * L.k * L.k : (size_t)L.k * L.k;
        if ((size_t)L.w_off + (size_t)L.cout * L.cin * taps > n_floats) return false;
        const bool last = is_out[L.dst] != 0;
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
int model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob, size_t n_floats,
               const std::vector<int>& preset_chan, tpz_model** out) {
    if (model_load_one(ctx, layers, n_layers, h_blob, n_floats, preset_chan, out)) return 1;
    tpz_model* m = *out;
    if (preset_chan.size() != 1 || m->dbg.no_widen || ctx->exact || m->n_conv_split == m->n_conv) return 0;
    // some layer has no 2xf16 kernel at the widths as given: try the zero-padded program, keep whichever covers more layers
    std::vector<tpz_layer> wl;
    std::vector<float> wb;
    if (!widen_program(layers, n_layers, h_blob, n_floats, wl, wb)) return 0;
    tpz_model* mw = nullptr;
    const std::string err_ctx = ctx->err, err_global = last_global_error();
    if (model_load_one(ctx, wl.data(), n_layers, wb.data(), wb.size(), preset_chan, &mw)) {
        restore_errors(ctx, err_ctx, err_global);      // the plain model is kept and the load succeeds: no stale error text
        return 0;
    }
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
    m->dbg = debug_env();         // the A/B switches of THIS load (all off without TPZ_DEBUG=1)
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
