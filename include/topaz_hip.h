/* libtopaz_hip.so -- C-ABI of the MI355X (gfx950) hot path of tbepler/topaz.
 *
 * The reference has no FFI layer: its boundary is the Python callable
 *     model(x: Tensor[N,1,H,W]) -> Tensor[N,1,H,W]          (nn.Module.__call__)
 * invoked from topaz/extract.py:249, topaz/model/utils.py:122, topaz/predict.py:26,
 * topaz/denoise.py:293, plus the pure functions non_maximum_suppression
 * (topaz/algorithms.py:25) / non_maximum_suppression_3d (:66) and Denoise.denoise
 * (topaz/denoise.py:328) / Denoise3D.denoise (:340).  Each entry point below names the
 * reference interface it replaces.  The Python package topaz_amd binds these with ctypes
 * (topaz_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; tpz_last_error(ctx) gives the text.
 *   - plain pointers and sizes only.  Pointers named d_* are DEVICE pointers (HBM of the
 *     ctx's GPU); h_* are host pointers.  fp32 everywhere, tensors are dense, C-order
 *     [C][D][H][W] (D omitted in 2-D), batch entries are processed one after another.
 *   - one ctx per process/GPU; calls on a ctx are serialised on its HIP stream
 *     (tpz_ctx_set_stream lets the caller supply e.g. torch's current stream).
 *     Calls are asynchronous w.r.t. the host unless stated; tpz_ctx_sync waits.
 *   - the library owns its workspaces (grow-only), the caller owns in/out buffers.
 */
#ifndef TOPAZ_HIP_H
#define TOPAZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tpz_ctx tpz_ctx;
typedef struct tpz_model tpz_model;

/* ---- layer program: the manifest a model is loaded from ------------------------------
 * A model is a list of ops over tensor "slots".  Slot 0 is the network input (1 channel);
 * the dst slot of the last layer is the network output.  The host packer
 * (topaz_amd/model/pack.py) emits this from a state_dict / unpickled module; it mirrors
 * the module graphs of topaz/model/features/resnet.py:243-251,185-202,
 * topaz/model/features/basic.py:98-111, topaz/model/classifier.py:64-66 and
 * topaz/denoising/models.py:130-175,515-562.                                              */
enum {
    TPZ_OP_CONV = 1,
    TPZ_OP_MAXPOOL2 = 2,     /* MaxPool(2) with floor (the U-Net encoders, denoising/models.py:81-97) */
    TPZ_OP_MAXPOOL = 3       /* k^dims max over a window dilated by `dil`, stride 1, no padding: the FILLED form of
                                MaxPool(3, stride = 2) in ResNet6 and the --pooling max ResNets (resnet.py:10-47,254-339) */
};

typedef struct tpz_layer {
    int32_t op;        /* TPZ_OP_* */
    int32_t dims;      /* 2 or 3 */
    int32_t src;       /* input slot */
    int32_t src2;      /* -1, or slot concatenated AFTER src on the channel axis; src is first
                          nearest-upsampled to src2's size (F.interpolate(mode='nearest') + torch.cat,
                          denoising/models.py:140-171) */
    int32_t dst;       /* output slot */
    int32_t cin;       /* total input channels (src [+ src2]) */
    int32_t cout;
    int32_t k;         /* cubic kernel size */
    int32_t dil;       /* dilation (the "filled" stride->dilation rewrite, resnet.py:87-92,153-164) */
    int32_t pad;       /* zero padding on every side */
    float slope;       /* activation y = v > 0 ? v : slope*v : 0 = ReLU, 0.1 = LeakyReLU, 1 = none, else PReLU */
    int64_t w_off;     /* offset (floats) of the [cout][cin][k(][k)][k] weights in the blob */
    int64_t b_off;     /* offset of the [cout] bias, -1 = none */
    int32_t res;       /* -1, or slot added before the activation (ResidA skip, resnet.py:185-202) */
    int32_t res_crop;  /* the skip is centre-cropped by this many pixels per side */
    int64_t post_scale_off; /* -1, or [cout] scale of the eval-BN affine applied AFTER the add (bn1) */
    int64_t post_shift_off;
    int32_t head;      /* 1: fuse a following 1x1 conv cout->1 (classifier.py:29,65); dst then has 1 channel */
    int32_t reserved;
    int64_t head_w_off;/* [cout] */
    int64_t head_b_off;/* [1] */
} tpz_layer;

/* ---- context ------------------------------------------------------------------------- */
/* replaces topaz/cuda.py:16-32 set_device (no CPU fallback: errors are reported, never hidden).
 * Threading: a context, with the models loaded into it, is driven by ONE host thread at a time -- the staging ring is the
 * exception: its producer side (the h2d call and the pinned buffers) may run on a reader thread, as the CLI does.  One
 * process per GPU with one ctx is the tested configuration (topaz --gpus N); several contexts on one device driven by
 * several threads at once are not. */
int tpz_ctx_create(int device_id, tpz_ctx** out);
void tpz_ctx_destroy(tpz_ctx* ctx);
const char* tpz_last_error(tpz_ctx* ctx);            /* ctx may be NULL: last global error */
int tpz_ctx_set_stream(tpz_ctx* ctx, void* hip_stream); /* NULL = the ctx's own stream */
int tpz_ctx_sync(tpz_ctx* ctx);
const char* tpz_version(void);

/* ---- models -------------------------------------------------------------------------- */
/* replaces topaz/model/factory.py:33-64 load_model + model.eval(); model.fill(); model.cuda()
 * (extract.py:227-232) and topaz/denoising/models.py:581-625 load_model + .cuda() (denoise.py:248-262):
 * weights are packed into MFMA fragment order and copied to the device; the model owns them. */
int tpz_model_load(tpz_ctx* ctx, const tpz_layer* layers, int n_layers, const float* h_blob,
                   size_t n_floats, tpz_model** out);
void tpz_model_free(tpz_model* m);

/* replaces model(x) -- LinearClassifier.forward (classifier.py:48-66), UDenoiseNet.forward
 * (denoising/models.py:130-175), UDenoiseNet3D.forward (:515-562).
 * d_in [n][1][D][H][W] -> d_out [n][1][Do][Ho][Wo]; D = 1 for 2-D models.
 * Arithmetic path per image: the 2xf16 kernels where the model has them, re-run on the fp32 kernels when an activation left
 * the f16 range; a 3-D volume whose widest activation exceeds 4 GiB per split half (32-bit offsets of the plane-stacked
 * kernels: e.g. 128 channels x more than 256^3 voxels) goes to the fp32 kernels directly -- or is tiled by the caller, as
 * classify_patches does.  A 2-D image above ~11 500^2 pixels is refused with a message: process it in patches. */
int tpz_model_forward(tpz_model* m, const float* d_in, int n, int D, int H, int W, float* d_out);
/* output size of the model for a given input size */
int tpz_model_out_shape(tpz_model* m, int D, int H, int W, int* Do, int* Ho, int* Wo);
/* channels of the model output (1 for every network of the reference; >1 only for partial programs) */
int tpz_model_out_channels(tpz_model* m, int* C);

/* ---- denoising ----------------------------------------------------------------------- */
/* replaces Denoise.denoise / denoise_patches / _denoise (topaz/denoise.py:274-332):
 * patch grid range(0,H,patch) x range(0,W,patch), each crop [i-pad, i+patch+pad) clipped to the
 * image is normalised by ITS OWN mean / unbiased std (torch.std), run through the net,
 * un-normalised and its centre pasted.  patch <= 0 or patch+pad >= max(H,W): whole image. */
int tpz_denoise_2d(tpz_model* m, const float* d_in, int H, int W, int patch, int pad, float* d_out);
/* replaces Denoise3D.denoise + PatchDataset (topaz/denoise.py:340-377,
 * topaz/denoising/datasets.py:412-468): global mean / population std (numpy), zero-filled
 * (patch+2*pad)^3 tiles normalised globally then per tile (unbiased), stitched.
 * patch < 1: the whole volume goes through _denoise. */
int tpz_denoise_3d(tpz_model* m, const float* d_in, int D, int H, int W, int patch, int pad, float* d_out);
/* One rank's share of a tomogram: only the tiles whose index (z-major over the ceil(n / patch)^3 grid, the order of
 * PatchDataset) is == shard (mod n_shards) are denoised and pasted; the rest of d_out is left untouched (zero it first
 * and sum the ranks' volumes).  The global mean / std are computed over the whole input by every rank -- a deterministic
 * reduction, so all ranks normalise identically without exchanging anything. */
int tpz_denoise_3d_shard(tpz_model* m, const float* d_in, int D, int H, int W, int patch, int pad, int shard, int n_shards,
                         float* d_out);
/* mean and std of n floats; unbiased != 0 -> divide by n-1 (torch.std), else by n (numpy.std)
 * (topaz/denoise.py:283,343,388).  h_mean_std[2] on the host; synchronises. */
int tpz_mean_std(tpz_ctx* ctx, const float* d_x, size_t n, int unbiased, float* h_mean_std);
/* 2-component Gaussian-mixture fit of pixel values, the model behind `topaz normalize` (topaz/stats.py:87-203
 * norm_fit + gmm_fit with a shared variance and a Beta(alpha, beta) prior on the mixing proportion): for each of
 * the n_init initialisations (mixing proportion pis[i], hard split at splits[i] = quantile(x, 1 - pis[i]);
 * pis[i] == 1 selects the single-Gaussian model) runs EM for at most num_iters iterations or until the
 * log-likelihood gains <= tol, and returns mean / std of the upper component, the MAP proportion and the final
 * log-likelihood (x scale, the sub-sampling factor).  d_x: n device floats.  One device pass per EM iteration,
 * statistics in fp64. */
int tpz_gmm_fit(tpz_ctx* ctx, const float* d_x, size_t n, const double* pis, const double* splits, int n_init,
                double alpha, double beta, double scale, int num_iters, double tol, double* mus, double* stds,
                double* pis_out, double* logps);
/* y = x*scale + shift  (the (x-mu)/std and std*y+mu steps of topaz/denoise.py:389,414) */
int tpz_affine(tpz_ctx* ctx, const float* d_x, size_t n, float scale, float shift, float* d_y);
/* y = (x - mean) / std evaluated as numpy does in `(mic - mu) / std` (topaz/denoise.py:389, :412 and commands/normalize):
 * fp32 subtraction, then an IEEE fp32 division -- exact low bits on raw-count micrographs (|mean| >> std); std == 0 yields the
 * reference's inf / nan. */
int tpz_normalize(tpz_ctx* ctx, const float* d_x, size_t n, float mean, float std, float* d_y);
/* 1->1 channel 2-D filter with zero "same" padding: GaussianDenoise / InvGaussianFilter /
 * AffineFilter.forward (topaz/filters.py:28-96).  h_w is the [k][k] kernel on the host. */
int tpz_filter_2d(tpz_ctx* ctx, const float* d_in, int H, int W, const float* h_w, int k, float bias,
                  float* d_out);

/* ---- non-maximum suppression ----------------------------------------------------------- */
/* replaces non_maximum_suppression(x, r, threshold) (topaz/algorithms.py:25-63), bit-identical
 * to the greedy loop including the clip-to-W quirk (offsets past the right edge suppress
 * column 0 of the next row).  Ties between equal scores are ordered by DESCENDING flat index
 * (what a stable argsort reversed gives; numpy's default argsort leaves tie order undefined).
 * Outputs (device): d_coords [cap][2] int32 (x, y), d_scores [cap] fp32, in descending score order.
 * *h_n receives the number of picks found (may exceed cap: then only cap rows were written
 * and the return code is non-zero).  Synchronises. */
int tpz_nms_2d(tpz_ctx* ctx, const float* d_score, int H, int W, int r, float threshold,
               int32_t* d_coords, float* d_scores, int cap, int* h_n);
/* replaces non_maximum_suppression_3d(x, r, scale, threshold) (topaz/algorithms.py:66-103):
 * flat-index deltas, no clipping (they wrap across rows/planes); d_coords [cap][3] (x, y, z). */
int tpz_nms_3d(tpz_ctx* ctx, const float* d_score, int D, int H, int W, int r, double scale,
               float threshold, int32_t* d_coords, float* d_scores, int cap, int* h_n);
/* (scale is a double: the reference forms r = scale * r in Python floats, and a lattice point at distance exactly r
 *  must stay inside the ball -- 0.7f * 10 is 6.9999998, 0.7 * 10 is 7.0.  The threshold is compared with fp32 scores
 *  as fp32, like numpy does for a float32 map.) */

/* ---- host buffers: staging ring and host-pointer entry points ----------------------------------------------
 * The reference's callers own numpy images (extract.py:234-251 loads one, scores it, hands the map on; denoise.py:463-488
 * likewise).  A tpz_stage is a ring of `depth` slots -- pinned host buffer + device buffer + events -- with its own copy
 * stream: the H2D copy of image i+1 and the D2H copy of result i-1 run under the kernels of image i.
 *   producer:  p = tpz_stage_host_ptr(st, k);  fill p (e.g. read the file into it);  tpz_stage_h2d(st, k, NULL, bytes)
 *              (or tpz_stage_h2d(st, k, h_src, bytes): h_src is first copied into the pinned buffer)
 *   consumer:  tpz_stage_acquire(st, k)  -- the ctx stream waits for the copy (no host wait);
 *              ... kernels reading tpz_stage_device_ptr(st, k) ...;  tpz_stage_release(st, k)
 *   results:   tpz_stage_d2h(st, k, d_src, bytes) queues device -> pinned slot k after the work already on the ctx stream;
 *              tpz_stage_wait(st, k) blocks the host until slot k's last copy has finished.
 * A slot is reused only after the kernels that read it were released (h2d waits for the release event). */
typedef struct tpz_stage tpz_stage;
int tpz_stage_create(tpz_ctx* ctx, size_t slot_bytes, int depth, tpz_stage** out);
void tpz_stage_free(tpz_stage* st);
void* tpz_stage_host_ptr(tpz_stage* st, int slot);
void* tpz_stage_device_ptr(tpz_stage* st, int slot);
int tpz_stage_h2d(tpz_stage* st, int slot, const void* h_src, size_t bytes);
int tpz_stage_acquire(tpz_stage* st, int slot);
int tpz_stage_release(tpz_stage* st, int slot);
int tpz_stage_d2h(tpz_stage* st, int slot, const void* d_src, size_t bytes);
int tpz_stage_wait(tpz_stage* st, int slot);
/* Host-pointer forms of the three calls of the path (synchronous; plain pageable or pinned memory), the signatures
 * SURVEY.md 8(b) sketches: model(x) on a numpy image (extract.py:247-249), Denoise.denoise (denoise.py:327-332),
 * non_maximum_suppression (algorithms.py:25-63).  Each stages through an internal ring owned by the ctx. */
int tpz_score_2d_host(tpz_model* m, const float* h_in, int H, int W, float* h_out_logits);
int tpz_denoise_2d_host(tpz_model* m, const float* h_in, int H, int W, int patch, int pad, float* h_out);
int tpz_nms_2d_host(tpz_ctx* ctx, const float* h_score, int H, int W, int r, float threshold, int32_t* h_coords,
                    float* h_scores, int cap, int* h_n);

/* ---- single ops (unit tests and the host-side pipelines) --------------------------------- */
/* one fused convolution: the op the layer program is made of.  h_w [cout][cin][k..], h_b [cout] or NULL.
 * d_in2 / d_res / h_post_* may be NULL.  in is [cin1][D1][H1][W1]; when d_in2 != NULL it is
 * nearest-upsampled to in2's [cin-cin1][D][H][W] and concatenated. */
int tpz_conv(tpz_ctx* ctx, int dims, const float* d_in, int cin1, int D1, int H1, int W1, const float* d_in2,
             int cin, int D, int H, int W, const float* h_w, const float* h_b, int cout, int k, int dil,
             int pad, float slope, const float* d_res, int res_crop, const float* h_post_scale,
             const float* h_post_shift, const float* h_head_w, float head_b, float* d_out);
int tpz_maxpool2(tpz_ctx* ctx, int dims, const float* d_in, int C, int D, int H, int W, float* d_out);
/* d_out[cols][rows] = d_in[rows][cols]; used by the truncated-DFT downsample (topaz/utils/image.py:38-61),
 * which runs as two GEMMs (tpz_conv with k = 1) around a transpose -- see topaz_amd/utils/image.py */
int tpz_transpose_2d(tpz_ctx* ctx, const float* d_in, int rows, int cols, float* d_out);

/* ---- introspection / measurement --------------------------------------------------------- */
/* ---- 2xf16 path.  Every convolution with a conv_split kernel -- the scoring networks (stem as a column kernel, dilated
 * 3x3 / 5x5 layers, folded 1x1 projections, fused head), the 2-D U-Nets / FCNN (encoders with fused max-pool, per-parity and
 * sub-pixel decoders, the 1-output-channel last conv as a column kernel) and the 3-D U-Net (plane-stacked) -- runs by default on
 * the f16 matrix cores with every fp32 operand carried as two f16 halves and three exact products accumulated in fp32
 * (topaz_amd/csrc/conv_split.h): fp32-level accuracy at several times the fp32-MFMA rate.  The rule per layer: it takes the 2xf16
 * path when a conv_split kernel exists for its shape (2-D and plane-stacked 3-D alike: the 3-D scoring networks run there too),
 * otherwise it stays on its fp32-MFMA kernel; layers with a PReLU slope > 1 always do.  An activation beyond the f16 range is detected on the device and that
 * image is re-run on the fp32 kernels, so results never depend on the range.  tpz_ctx_set_exact(ctx, 1) pins the
 * fp32 kernels.
 * tpz_model_split_stats: whether the model is eligible, images finished on the 2xf16 path, images re-run in fp32. */
int tpz_ctx_set_exact(tpz_ctx* ctx, int on);
/* Debug switches.  The library's A/B switches (TPZ_NO_ROI, TPZ_EXACT_FP32, TPZ_NO_LANES, ... -- the list is the DebugEnv struct of
 * topaz_amd/csrc/rt_internal.h) exist for tools/ and tests/; a variable left in a user's environment must not change how a job
 * computes, so the library reads them in ONE function and only when TPZ_DEBUG=1 is set as well.  tpz_debug_switches writes the
 * space-separated names of the switches that are in effect right now into buf ("" when none: always so without TPZ_DEBUG=1)
 * and returns their number.  Needs no device. */
int tpz_debug_switches(char* buf, int buf_len);
/* Range scaling of the scoring pass (tpz_model_forward of a program that ends in the linear head), on by default
 * (on = 0: off).  `topaz extract` scores micrographs as they come (topaz/extract.py:234-249 does not normalise), and
 * a raw-count image (mean 10^3 .. 10^4) would leave the f16 range in the very first layer and send the whole image to the fp32
 * kernels.  A scoring network is positively homogeneous in (input, biases) jointly -- convolutions, PReLU / ReLU, max-pools,
 * residual adds, eval-BN affines, the linear head --, so it is run on x * 2^-s with every bias-like vector scaled by 2^-s and its
 * logits multiplied by 2^s: exact (powers of two).  s >= 0 brings the 99.9 % quantile of |x| (from an exponent histogram taken on
 * the device, no host round trip) to ~8: the bulk of the image decides, so a hot pixel cannot push the rest towards the f16
 * subnormals -- it either still fits the f16 range or trips the overflow flag and the image is re-run in fp32.  A normalised image
 * has s = 0 and runs unchanged, bit for bit. */
int tpz_ctx_set_range(tpz_ctx* ctx, int on);
/* Patch lanes: tpz_denoise_2d / _3d enqueue the independent patches / tiles of an image alternately on two auxiliary
 * streams (own workspace each), so that one patch's small, latency-bound launches run under its neighbour's large ones.
 * On by default (on = 0 here: everything on the ctx stream, e.g. to time kernels in
 * isolation).  on = 2 .. 4: that many lanes; more than two measured no gain on the 4096^2 pipeline
 * (profiles/r03_lanes.txt).  Results are bit-identical either way. */
int tpz_ctx_set_lanes(tpz_ctx* ctx, int on);
/* Batched patches (2xf16 path): tpz_denoise_2d / _3d record the launches of up to n (<= 8) independent patches / tiles and issue
 * them layer by layer, the same layer of all n as ONE grid (conv_split_multi_kernel): the deep levels of a U-Net are 16-tile
 * launches on a 256-CU chip, ~400 of them per micrograph (denoise.py:299-323 runs the patches one after another).  Consecutive
 * batches alternate on the patch lanes above (a batch's elementwise launches and small grids run under the other batch's large
 * ones: profiles/r04_batch_lanes_ab.txt).  Default n = 8; n = 0: off -- single
 * patches then alternate on the lanes, as do the fp32 kernels (exact mode, overflow re-run) always.  Results are bit-identical
 * either way.
 * Every image of a batch has a workspace of its own and two batches are in flight: the batch is cut to what fits 90 % of the
 * free device memory (tpz_ctx_set_batch_memory(ctx, bytes): that many bytes instead; 0 = automatic), below 2 the pass falls
 * back to single patches on the lanes -- a large tile costs launches, not an out-of-memory error.
 * tpz_prof_launches: kernel launches the library has issued on this context (convolutions, elementwise, NMS sweeps excluded). */
int tpz_ctx_set_batch(tpz_ctx* ctx, int n);
int tpz_ctx_set_batch_memory(tpz_ctx* ctx, long long bytes);
long long tpz_prof_launches(tpz_ctx* ctx);
/* Patch windows: a patch of tpz_denoise_2d keeps only its centre (topaz/denoise.py:299-323: patch_size pixels of a
 * patch_size + 2*padding tile), so each layer computes only the rectangle of its tensor that those pixels depend on (the
 * U-Net's receptive field is ~230 pixels, the CLI's default padding 500).  The statistics of the normalisation are still the
 * whole padded patch's; every kept pixel is computed exactly as before -- the output is bit-identical with the switch off
 * (on = 0 here).  Both paths: the 2xf16 kernels and, under tpz_ctx_set_exact, the fp32 ones.
 * The tiles of tpz_denoise_3d (denoise.py:340-377: patch^3 voxels kept of a (patch + 2*padding)^3 tile, 1/8 at the CLI's 96 / 48)
 * are windowed the same way with boxes, on the 2xf16 kernels and (round 5) on the fp32 kernels of exact mode / an overflow
 * re-run: 3.5x on a 512x512x256 tomogram, bit-identical. */
int tpz_ctx_set_roi(tpz_ctx* ctx, int on);
/* The 3x3 32 -> 32 layers of the 32-unit detectors (resnet8_u32 / resnet16_u32: topaz/model/features/resnet.py:108-204 filled)
 * run on a persistent kernel that keeps the layer's packed weights in the LDS (csrc/conv_rw.h); on = 0 (with TPZ_DEBUG=1: TPZ_NO_RW=1 at load
 * time) sends them to the general 2xf16 tile again.  Same tensors either side; results agree to rounding (another summation
 * order), each within 1e-4 of the reference. */
int tpz_ctx_set_rw(tpz_ctx* ctx, int on);
/* Persistent workgroups of the 2xf16 convolutions (conv_split.h, MODE 4): a plain single-source layer with several tiles per
 * workgroup slot is launched as CUs x workgroups-per-CU workgroups that walk the tiles and fetch the first chunk of their next
 * tile under the last chunk of the current one.  mode 0: never, 1: large launches outside the patch lanes
 * (default), 2: every eligible launch, with `workgroups` of them (0: one grid slot each) -- the form the tests use to run
 * small images through many tiles per workgroup.  Results are bit-identical in every mode. */
int tpz_ctx_set_persist(tpz_ctx* ctx, int mode, int workgroups);
/* Patch raster of the large 8-wave launches (conv_split.h, xcd_swizzle 2): the 32 tiles an XCD's CUs hold at a time form an
 * 8 x 4 block of neighbouring tiles (rows taken phase-major for dilated layers), so that their halos overlap in that XCD's L2,
 * instead of a row-major run that shares columns only.  On by default (on = 0: row-major runs).  Same
 * arithmetic per tile: results are bit-identical. */
int tpz_ctx_set_raster(tpz_ctx* ctx, int on);
/* Internal tiling of tpz_model_forward: a 2-D image of more than limit_px pixels (default 40 Mi: beyond a 6400^2 frame) through a
 * size-preserving scoring network is scored in tile x tile tiles (default 4096), each computed on the tile grown by the network's
 * receptive halo and stitched -- where the reference scores any image that fits memory (topaz/extract.py:247-249), the kernels
 * address 32-bit byte offsets per chunk of cells (< ~11 500^2 pixels) and a whole-image pass would hold every activation at full
 * size (an 11 520 x 8 184 super-resolution frame: 146 GB).  Bit-identical to the whole-image pass. */
int tpz_ctx_set_tiling(tpz_ctx* ctx, long long limit_px, int tile);
int tpz_model_split_stats(tpz_model* m, int* eligible, long long* split_runs, long long* fp32_reruns);
/* Coverage of the 2xf16 path: convolution layers of the model, how many of them have a 2xf16 kernel, and the others as text
 * ("#layer KxK dD cin->cout, ...").  `eligible` above is true as soon as ONE layer has: the rest of a mixed program runs on its
 * fp32 kernels (correct, converted either side, several times slower) -- topaz_amd warns once per model when that happens. */
int tpz_model_split_layers(tpz_model* m, int* n_conv, int* n_split, char* off_path, int off_path_len);
/* One 2-D convolution on the 2xf16 kernels with fp32 [C][H][W] tensors at the boundary (converted on the device):
 * unit-test / interop entry; arguments as tpz_conv (single source).  *overflow = 1 when a result left the f16 range. */
int tpz_conv_split_2d(tpz_ctx* ctx, const float* d_in, int cin, int H, int W, const float* h_w, const float* h_b,
                      int cout, int k, int dil, int pad, float slope, const float* d_res, int res_crop,
                      const float* h_post_scale, const float* h_post_shift, const float* h_head_w, float head_b,
                      float* d_out, int* overflow);

/* time (ms, HIP events on the ctx stream) and launch count of the kernels of one class since
 * the last reset.  cls: 0 = conv_mfma, 1 = conv_direct, 2 = elementwise, 3 = nms.
 * Timing is only collected while enabled (it adds two event records per timed launch): on = 1 times every launch,
 * on = 2 only the convolution launches of >= 20 GFLOP (cheap enough to stay enabled during a benchmark's timed steps). */
int tpz_prof_enable(tpz_ctx* ctx, int on);
int tpz_prof_reset(tpz_ctx* ctx);
int tpz_prof_get(tpz_ctx* ctx, int cls, double* ms, long long* launches, double* flops);
/* the conv_mfma instantiation with the largest accumulated time since the last reset: total ms, launch count,
 * algorithmic FLOP of those launches and its template parameters as text (matches the rocprofv3 kernel name) */
int tpz_prof_get_dominant(tpz_ctx* ctx, double* ms, long long* launches, double* flops, char* name, int name_len);
/* the same for the rank-th instantiation by accumulated time (rank 0 = the dominant one); an empty name and zeros
 * once rank runs past the instantiations that were launched */
int tpz_prof_get_kernel(tpz_ctx* ctx, int rank, double* ms, long long* launches, double* flops, char* name,
                        int name_len);
/* algorithmic HBM bytes (inputs with halo + weights read once, outputs written once) of the launches behind
 * tpz_prof_get_kernel(rank): the bandwidth of the kernels that are HBM-bound rather than MFMA-bound */
int tpz_prof_get_kernel_bytes(tpz_ctx* ctx, int rank, double* bytes);
/* The matrix-pipe rate this board sustains (diagnostic, no reference counterpart): a register-resident loop of
 * v_mfma_f32_16x16x32_f16 -- the instruction of the 2xf16 kernels -- on every SIMD for about `ms` milliseconds, operands
 * uniform in [-1, 1] (zero_operands = 0) or all zero (1).  *tflops = dense f16 TFLOP/s (3 such products make one fp32-equivalent
 * multiply-add on the 2xf16 path); *clock_ratio = s_memtime / s_memrealtime ticks over the loop (proportional to the shader
 * clock; may be NULL).  With full-entropy operands the power management holds the clock well below the 2.4 GHz the dense peak
 * is quoted at: bench.py reports the dominant kernel against both. */
int tpz_prof_mfma_sustained(tpz_ctx* ctx, int ms, int zero_operands, double* tflops, double* clock_ratio);

/* ---- pick-table text (host only) ------------------------------------------------------------------------------------
 * replaces the per-pick f-string of topaz/extract.py:341-354: n rows `image_name \t x \t y [\t z] \t score \n` into `out`
 * (capacity `cap` bytes; strlen(image_name) + 80 per row is always enough).  coords: int32, `coord_stride` ints per row of
 * which the first `dims` are written; a float32 score prints as Python prints float(score): the shortest digits that
 * round-trip the float64 value (repr).  Returns the bytes written, -1 on bad arguments, -2 when `out` is too small. */
long long tpz_format_picks(const char* image_name, const int32_t* coords, int coord_stride, int dims, const float* scores,
                           long long n, char* out, long long cap);

#ifdef __cplusplus
}
#endif
#endif /* TOPAZ_HIP_H */
