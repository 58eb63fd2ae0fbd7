// launch helpers implemented in kernels_misc.hip and nms.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.h"

namespace tpz {

hipError_t launch_conv_direct(const ConvArgs& a, const float* d_w, int K, int KZ, int dil, hipStream_t s);
hipError_t launch_maxpool2(const float* in, float* out, int C, int D, int H, int W, int dims, hipStream_t s);
hipError_t launch_meanstd(const float* x, int D, int H, int W, long long ps, int pitch, int unbiased, int mode,
                          const float* d_g, double* d_part, int part_blocks, float* d_out, hipStream_t s);
hipError_t launch_gmm_pass(const float* x, size_t n, int mode, const double* d_par, double* d_part, int part_blocks,
                           double* d_out, hipStream_t s);
hipError_t launch_affine(const float* x, float* y, size_t n, float scale, float shift, hipStream_t s);
hipError_t launch_normalize(const float* x, float* y, size_t n, float mu, float sd, hipStream_t s);     // y = (x - mu) / sd
hipError_t launch_affine_dev(const float* x, int D, int H, int W, long long ps, int pitch, const float* d_p, float* y,
                             hipStream_t s);
hipError_t launch_transpose(const float* in, float* out, int R, int Cc, hipStream_t s);
hipError_t launch_maxpool2_split(const void* in, void* out, int C, int D, int H, int W, int dims, hipStream_t s);
hipError_t launch_maxpoolz_split(const float* in, float* out, int C, int D, int H, int W, hipStream_t s);   // z pairs only (split cells)
hipError_t launch_maxpoolk(const void* in, void* out, int C, int D, int H, int W, int k, int dil, int dims, bool split,
                           hipStream_t s);
hipError_t launch_shiftx_split(const float* in, void* out, int K, int pad, size_t rows, int W, int Wo, unsigned* flag,
                               hipStream_t s, size_t r0 = 0, size_t r1 = (size_t)-1, int x0 = 0, int x1 = 0x7fffffff);
hipError_t launch_shiftsum(const float* Y, float* out, int K, size_t rows, int W, int Wp, float bias, const float* nrm,
                           int norm_out, hipStream_t s, size_t y0 = 0, size_t y1 = (size_t)-1, int x0 = 0, int x1 = 0x7fffffff,
                           const float* res = nullptr, int Hp = 0, int z0 = 0, int z1 = 1);   // Hp > 0: planes [z0, z1) of Hp rows
// 1-output-channel k^dims conv over split cells on the vector ALUs (K = 3 or 5; KZ = K for a 3-D conv, else 1), fused with bias,
// same-size residual and un-normalisation; wt = [KZ][cells][kx][ky][8] fp32
hipError_t launch_conv_cout1_split(const void* in, const float* wt, float* out, const float* res, const float* nrm, int norm_out,
                                   float bias, int cells, int K, int KZ, int D, int H, int W, int z0, int z1, int y0, int y1,
                                   int x0, int x1, hipStream_t s);
hipError_t launch_s2d_split(const float* in, void* out, int d, int h, int w, int H, int W, int dims, unsigned* flag,
                            hipStream_t s);
hipError_t launch_to_split(const float* in, void* out, int C, int H, int W, unsigned* flag, hipStream_t s);
hipError_t launch_from_split(const void* in, float* out, int C, int H, int W, hipStream_t s);
hipError_t launch_copy_box(const float* src, long long sps, int spitch, float* dst, long long dps, int dpitch, int bd,
                           int bh, int bw, hipStream_t s);
// Range scaling of a scoring pass (rt_forward.hip tpz_model_forward): rng[0] = 2^-s, rng[1] = 0, rng[2] = 2^s, rng[3] = s with s >= 0
// chosen from the exponent histogram of the image (its 99.9 % quantile of |x| brought to ~2^3); dst[i] = src[i] * 2^-s for the
// model's bias-like vectors.  `hist` = 256 zeroed words (left zeroed).
hipError_t launch_range_fit(const float* x, size_t n, unsigned* hist, float* rng, const float* src, float* dst, size_t n_vec,
                            hipStream_t s);
// y[i] = y[i] * rng[2] + add (skipped on the device when that is the identity)
hipError_t launch_unscale(float* y, size_t n, const float* rng, float add, hipStream_t s);
// diag.hip: n_wg 4-wave workgroups each issue 8 * iters v_mfma_f32_16x16x32_f16 per wave on operands read from src (4096 x 16 B);
// ticks[0] = s_memtime, ticks[1] = s_memrealtime (100 MHz) ticks of wave 0's loop
hipError_t launch_mfma_spin(const void* src, float* out, int n_wg, int iters, unsigned long long* ticks, hipStream_t s);
hipError_t launch_extract_tile3d(const float* tomo, int D, int H, int W, int i0, int j0, int k0, int d,
                                 const float* d_g, float* tile, hipStream_t s);

hipError_t nms_mark(const float* score, size_t n, float thr, uint8_t* status, uint32_t* cand, unsigned int* counters,
                    hipStream_t s);
hipError_t nms2d_sweep(const float* score, int H, int W, const int* near_cells, int n_near, const int* cells, int ncells,
                       uint8_t* status, const uint32_t* list_in, uint32_t* list_out, uint32_t* list_ver, unsigned int* cnt,
                       unsigned int* cnt_ver, unsigned int* snap, uint64_t* keys, unsigned int* npicks, size_t hint,
                       hipStream_t s);
hipError_t nms3d_sweep(const float* score, long long n, const int* near_deltas, int n_near, const int* deltas, int ndelta,
                       uint8_t* status, const uint32_t* list_in, uint32_t* list_out, uint32_t* list_ver, unsigned int* cnt,
                       unsigned int* cnt_ver, unsigned int* snap, uint64_t* keys, unsigned int* npicks, size_t hint,
                       hipStream_t s);
hipError_t fill_u64(uint64_t* p, size_t lo, size_t hi, uint64_t v, hipStream_t s);
hipError_t bitonic_sort_desc(uint64_t* keys, size_t npow2, hipStream_t s);
hipError_t nms_write(const uint64_t* keys, unsigned int n, const float* score, int H, int W, int dims,
                     int32_t* coords, float* out_scores, hipStream_t s);

}  // namespace tpz
