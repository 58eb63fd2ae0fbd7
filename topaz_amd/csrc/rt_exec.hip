// The layer-program executor: launchers, per-layer drivers, windows (need_regions), run_program.
#include "rt_internal.h"

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
    // (the residual plane is addressed with 32-bit offsets too, and with res_crop > 0 it is larger than the input plane)
    if (a.res && (size_t)a.cells_out * a.Hres * a.Wres * 16 >= ((size_t)1 << 32) - 16)
        return fail(ctx, "residual tensor too large for one launch (%d x %d): process the image in patches", a.Hres, a.Wres);
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
int run_conv_split(tpz_ctx* ctx, const LayerRT& rt, const Slot& s1, const Slot* sres, Slot& dst, const Slot* s2, bool pooled,
                   const Slot* fold) {
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
    a.issuer_half = ks.WAVES == 8 && ks.MT >= 96 && !ctx->dbg.no_issuer;   // -3 .. -4 % on the 128-channel tiles, nothing at 64 (tools/split_ablate.hip)
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
        if (m->ctx->dbg.trace_host) fprintf(stderr, "[tpz host] need_regions: program left whole (check %d)\n", why);
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
    if (m->ctx->dbg.trace_host)
        for (int i = 0; i < nl; ++i) {
            const tpz_layer& L = m->layers[i].L;
            const Rect& r = need[L.dst];
            fprintf(stderr, "[tpz host] need_regions: layer %d op %d -> slot %d: z [%d, %d) of %d, y [%d, %d) of %d, x [%d, %d) of %d\n", i,
                    (int)L.op, L.dst, r.z0, r.z1, Ds[L.dst], r.y0, r.y1, Hs[L.dst], r.x0, r.x1, Ws[L.dst]);
        }
    return need;
}

int run_program(tpz_model* m, std::vector<Slot>& slots, float* d_out, const float* d_nrm, bool split, const Rect* keep) {
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

