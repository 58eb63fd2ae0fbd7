// Greedy non-maximum suppression as a parallel fix-point, bit-identical to the sequential
// reference loop (topaz/algorithms.py:25-63 2-D, :66-103 3-D).
//
// Greedy: visit pixels in descending score; a pixel above the threshold is a pick iff no
// earlier pick's suppression set contains it.  Equivalently, with "priority" = visiting order:
//     p is SELECTED   iff every higher-priority candidate q with p in Supp(q) is SUPPRESSED
//     p is SUPPRESSED iff some higher-priority candidate q with p in Supp(q) is SELECTED
// which is iterated to its (unique) fix-point.  States only move UNDECIDED -> final and a final
// state is always correct when written, so kernels may read states other threads are updating.
//
// Supp(q) in 2-D (algorithms.py:58-61): flat = clip(qy+ii,0,H)*W + clip(qx+jj,0,W), ii^2+jj^2<=r^2.
//   The upper clip bound is W (not W-1): offsets past the right edge land on flat index
//   (y'+1)*W + 0, i.e. COLUMN 0 OF THE NEXT ROW is suppressed; offsets past the bottom fall
//   outside the array.  Low-side clipping only re-adds pixels the unclipped disk already holds.
//   => p=(py,px) is in Supp(q) iff (py-qy)^2+(px-qx)^2 <= r^2, or
//      px==0, py>=1 and (py-1-qy)^2 + (W-qx)^2 <= r^2.
// Supp(q) in 3-D (algorithms.py:78-79,100-101): {q + ii*zs + jj*ys + kk}, no clipping at all
//   (deltas wrap across rows/planes); the delta set is symmetric.
// Priority: descending score, ties by DESCENDING flat index (stable argsort reversed);
//   -0.0 == +0.0, NaN sorts above everything and passes `A[i] <= threshold` like in numpy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels_misc.h"

namespace tpz {

enum : uint8_t { ST_NONE = 0, ST_UNDECIDED = 1, ST_SELECTED = 2, ST_SUPPRESSED = 3 };

__device__ __forceinline__ uint32_t orderable(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;   // any NaN sorts above everything (numpy puts NaN last)
    if (u == 0x80000000u) u = 0u;                    // -0.0 compares equal to +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t prio_key(float f, uint32_t idx) {
    return ((uint64_t)orderable(f) << 32) | idx;
}

__global__ __launch_bounds__(256) void nms_mark_kernel(const float* __restrict__ score, size_t n, float thr,
                                                       uint8_t* __restrict__ status, uint32_t* __restrict__ cand,
                                                       unsigned int* __restrict__ counters) {
    // 1024 elements per block iteration; one atomic per iteration (wave ballots + LDS prefix over the 4 waves)
    __shared__ unsigned int wave_cnt[4];
    __shared__ unsigned int block_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += (size_t)gridDim.x * 1024) {
        bool is_c[4];
        unsigned int mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            is_c[k] = false;
            if (i < n) {
                const float s = score[i];
                is_c[k] = !(s <= thr);
                status[i] = is_c[k] ? ST_UNDECIDED : ST_NONE;
            }
            mine += is_c[k] ? 1u : 0u;
        }
        // exclusive prefix of `mine` inside the wave
        unsigned int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_cnt[wv] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            block_base = tot ? atomicAdd(&counters[0], tot) : 0u;
        }
        __syncthreads();
        unsigned int pos = block_base + (incl - mine);
        for (int w = 0; w < wv; ++w) pos += wave_cnt[w];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (is_c[k]) cand[pos++] = (uint32_t)(base + (size_t)k * 256 + threadIdx.x);
        __syncthreads();
    }
}

// One relaxation sweep over the list of still-undecided candidates (2-D).
//   pull: an undecided p looks for ANY higher-priority candidate among its possible suppressors that is not
//         (yet) SUPPRESSED; the first one found blocks p this sweep (if that neighbour is SELECTED its push
//         will suppress p, if it is UNDECIDED p has to wait).  Rows nearest first and, inside a row, columns
//         nearest first: in a dense candidate field an immediate neighbour is higher half of the time, so
//         most candidates stop after one or two probes.
//   select: a p with no such neighbour is SELECTED; its sort key goes straight to the pick list.  (Two kernels: a cheap
//         per-thread filter over the nearest cells, then one wave per survivor over the whole set -- a thread scanning the
//         ~pi r^2 cells of a local maximum alone kept its wave busy for hundreds of dependent probes.)
//   push (nms2d_push_kernel, after the sweep): every new pick marks the lower-priority candidates of Supp(p)
//         SUPPRESSED (including the column-0-of-the-next-row cells of the right-edge quirk).
//   Candidates that stay undecided are appended to the next sweep's list (wave-aggregated atomics), so sweep k
//   only touches what sweep k-1 left over.  The list lengths live on the device (cnt[0] in, cnt[1] out): the
//   host queues several sweeps back to back and reads a counter only once per batch.
__device__ __forceinline__ bool nms_higher(const float* __restrict__ score, const uint8_t* status, uint32_t q, uint64_t kp) {
    const uint8_t st = status[q];
    if (st == ST_NONE || st == ST_SUPPRESSED) return false;
    return prio_key(score[q], q) > kp;
}
// ---- sweep, phase A (one THREAD per still-undecided candidate): probe only the nearest cells of the suppressor set (the
// 5 x 5 neighbourhood clipped to the disk / the +-1 cube clipped to the ball, nearest first).  Every candidate that is still
// undecided goes to the next sweep's list; one that is also the best of its neighbourhood goes to the verify list.  In a dense
// candidate field > 90 % of the candidates are blocked here after one or two probes.
// List appends are aggregated per workgroup iteration (1024 candidates -> one atomic per list): device-scope atomics on a
// single address retire at ~10 ns each on this part, so one atomic per wave (or per candidate) would dominate the sweep.
struct BlockAppend {
    unsigned int wave_cnt[2][4];
    unsigned int base[2];
};
// each thread holds up to 4 (value, flag0, flag1) entries; appends value to list0 where flag0 and to list1 where flag1
__device__ __forceinline__ void block_append2(BlockAppend& sh, const uint32_t (&v)[4], const bool (&f0)[4], const bool (&f1)[4],
                                              uint32_t* __restrict__ list0, unsigned int* cnt0, uint32_t* __restrict__ list1,
                                              unsigned int* cnt1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int mine[2] = {0, 0}, incl[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mine[0] += f0[k] ? 1u : 0u; mine[1] += f1[k] ? 1u : 0u; }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        incl[l] = mine[l];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl[l], o, 64);
            if (lane >= o) incl[l] += t;
        }
        if (lane == 63) sh.wave_cnt[l][wv] = incl[l];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int l = threadIdx.x;
        const unsigned int tot = sh.wave_cnt[l][0] + sh.wave_cnt[l][1] + sh.wave_cnt[l][2] + sh.wave_cnt[l][3];
        sh.base[l] = tot ? atomicAdd(l == 0 ? cnt0 : cnt1, tot) : 0u;
    }
    __syncthreads();
    unsigned int pos0 = sh.base[0] + (incl[0] - mine[0]), pos1 = sh.base[1] + (incl[1] - mine[1]);
    for (int w = 0; w < wv; ++w) { pos0 += sh.wave_cnt[0][w]; pos1 += sh.wave_cnt[1][w]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (f0[k]) list0[pos0++] = v[k];
        if (f1[k]) list1[pos1++] = v[k];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void nms2d_filter_kernel(const float* __restrict__ score, int H, int W,
                                                           const int* __restrict__ near_cells, int n_near, const uint8_t* status,
                                                           const uint32_t* __restrict__ list_in, uint32_t* __restrict__ list_out,
                                                           uint32_t* __restrict__ list_ver, unsigned int* cnt, unsigned int* cnt_ver) {
    __shared__ BlockAppend sh;
    const unsigned int n_in = cnt[0];
    for (unsigned int base = blockIdx.x * 1024; base < n_in; base += gridDim.x * 1024) {    // whole workgroups iterate together
        uint32_t p[4];
        bool undecided[4], survivor[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int c = base + k * 256 + threadIdx.x;
            p[k] = 0;
            undecided[k] = survivor[k] = false;
            if (c < n_in) {
                p[k] = list_in[c];
                undecided[k] = status[p[k]] == ST_UNDECIDED;
            }
            if (undecided[k]) {
                const int py = (int)(p[k] / (uint32_t)W), px = (int)(p[k] % (uint32_t)W);
                const uint64_t kp = prio_key(score[p[k]], p[k]);
                bool blocked = false;
                for (int t = 0; t < n_near && !blocked; ++t) {
                    const int code = near_cells[t];
                    const int qy = py + (code >> 16), qx = px + (code & 0xffff) - 32768;
                    if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W)
                        blocked = nms_higher(score, status, (uint32_t)((size_t)qy * W + qx), kp);
                }
                survivor[k] = !blocked;
            }
        }
        block_append2(sh, p, undecided, survivor, list_out, cnt + 1, list_ver, cnt_ver);
    }
}
__global__ __launch_bounds__(256) void nms3d_filter_kernel(const float* __restrict__ score, long long n,
                                                           const int* __restrict__ near_deltas, int n_near, const uint8_t* status,
                                                           const uint32_t* __restrict__ list_in, uint32_t* __restrict__ list_out,
                                                           uint32_t* __restrict__ list_ver, unsigned int* cnt, unsigned int* cnt_ver) {
    __shared__ BlockAppend sh;
    const unsigned int n_in = cnt[0];
    for (unsigned int base = blockIdx.x * 1024; base < n_in; base += gridDim.x * 1024) {
        uint32_t p[4];
        bool undecided[4], survivor[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned int c = base + k * 256 + threadIdx.x;
            p[k] = 0;
            undecided[k] = survivor[k] = false;
            if (c < n_in) {
                p[k] = list_in[c];
                undecided[k] = status[p[k]] == ST_UNDECIDED;
            }
            if (undecided[k]) {
                const uint64_t kp = prio_key(score[p[k]], p[k]);
                bool blocked = false;
                for (int t = 0; t < n_near && !blocked; ++t) {
                    const long long q = (long long)p[k] + near_deltas[t];
                    if (q >= 0 && q < n) blocked = nms_higher(score, status, (uint32_t)q, kp);
                }
                survivor[k] = !blocked;
            }
        }
        block_append2(sh, p, undecided, survivor, list_out, cnt + 1, list_ver, cnt_ver);
    }
}

// the (at most four) winners of a workgroup iteration: SELECTED, keys appended with one atomic for the workgroup
__device__ __forceinline__ void winners_append(unsigned int* sh, bool won, uint32_t p, uint64_t kp, uint8_t* status,
                                               uint64_t* __restrict__ keys, unsigned int* npicks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = won ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int tot = sh[0] + sh[1] + sh[2] + sh[3];
        sh[4] = tot ? atomicAdd(npicks, tot) : 0u;
    }
    __syncthreads();
    if (lane == 0 && won) {
        unsigned int pos = sh[4];
        for (int w = 0; w < wv; ++w) pos += sh[w];
        status[p] = ST_SELECTED;
        keys[pos] = kp;
    }
    __syncthreads();
}

// ---- sweep, phase B (one WAVE per candidate that survived phase A): the lanes stride over the whole suppressor set and
// the wave stops at the first 64-cell slice that holds a higher-priority candidate not (yet) SUPPRESSED.  No such cell:
// the candidate is SELECTED and its sort key goes straight to the pick list; otherwise it stays undecided (it already is
// on the next sweep's list).
// 2-D suppressor set of p = (py, px): the disk around p, plus -- when px == 0 and py >= 1 -- the cells (py - 1 + dy, W - dx),
// dx >= 1, of the right-edge wrap quirk.
__global__ __launch_bounds__(256) void nms2d_verify_kernel(const float* __restrict__ score, int H, int W,
                                                           const int* __restrict__ cells, int ncells, uint8_t* status,
                                                           const uint32_t* __restrict__ list_ver, const unsigned int* __restrict__ cnt_ver,
                                                           uint64_t* __restrict__ keys, unsigned int* npicks) {
    __shared__ unsigned int sh_win[8];
    const unsigned int n_ver = cnt_ver[0];
    const int lane = threadIdx.x & 63;
    for (unsigned int i0 = blockIdx.x * 4; i0 < n_ver; i0 += gridDim.x * 4) {      // whole workgroups iterate together
        const unsigned int i = i0 + (threadIdx.x >> 6);
        const uint32_t p = list_ver[i < n_ver ? i : n_ver - 1];
        const int py = (int)(p / (uint32_t)W), px = (int)(p % (uint32_t)W);
        const uint64_t kp = prio_key(score[p], p);
        bool blocked = i >= n_ver;                             // (a wave past the end of the list: nothing to decide)
        for (int c0 = 0; c0 < ncells && !blocked; c0 += 64) {
            bool hit = false;
            const int c = c0 + lane;
            if (c < ncells) {
                const int code = cells[c];
                const int qy = py + (code >> 16), qx = px + (code & 0xffff) - 32768;
                if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W && !(qy == py && qx == px))
                    hit = nms_higher(score, status, (uint32_t)((size_t)qy * W + qx), kp);
            }
            blocked = __any(hit);
        }
        if (!blocked && px == 0 && py >= 1) {
            for (int c0 = 0; c0 < ncells && !blocked; c0 += 64) {
                bool hit = false;
                const int c = c0 + lane;
                if (c < ncells) {
                    const int code = cells[c];
                    const int dy = code >> 16, dx = (code & 0xffff) - 32768;
                    const int qy = py - 1 + dy, qx = W - dx;
                    if (dx >= 1 && (unsigned)qy < (unsigned)H && qx >= 0)
                        hit = nms_higher(score, status, (uint32_t)((size_t)qy * W + qx), kp);
                }
                blocked = __any(hit);
            }
        }
        winners_append(sh_win, !blocked, p, kp, status, keys, npicks);
    }
}
__global__ __launch_bounds__(256) void nms3d_verify_kernel(const float* __restrict__ score, long long n,
                                                           const int* __restrict__ deltas, int ndelta, uint8_t* status,
                                                           const uint32_t* __restrict__ list_ver, const unsigned int* __restrict__ cnt_ver,
                                                           uint64_t* __restrict__ keys, unsigned int* npicks) {
    __shared__ unsigned int sh_win[8];
    const unsigned int n_ver = cnt_ver[0];
    const int lane = threadIdx.x & 63;
    for (unsigned int i0 = blockIdx.x * 4; i0 < n_ver; i0 += gridDim.x * 4) {
        const unsigned int i = i0 + (threadIdx.x >> 6);
        const uint32_t p = list_ver[i < n_ver ? i : n_ver - 1];
        const uint64_t kp = prio_key(score[p], p);
        bool blocked = i >= n_ver;
        for (int d0 = 0; d0 < ndelta && !blocked; d0 += 64) {
            bool hit = false;
            const int d = d0 + lane;
            if (d < ndelta && deltas[d] != 0) {
                const long long q = (long long)p + deltas[d];
                if (q >= 0 && q < n) hit = nms_higher(score, status, (uint32_t)q, kp);
            }
            blocked = __any(hit);
        }
        winners_append(sh_win, !blocked, p, kp, status, keys, npicks);
    }
}

// The picks a sweep selected (keys[snap[0] .. snap[1])) suppress their lower-priority neighbours: one WAVE per pick, the
// lanes striding over the suppression set (a thread-per-pick loop over the ~pi r^2 cells left 63 lanes of its wave idle
// for hundreds of dependent iterations: 1.9 of the 4.4 ms of a 4096^2 map).  2-D: `cells` lists the (dy, dx) pairs of the
// disk as dy * 65536 + (dx + 32768); a cell past the right edge is column 0 of the next row (the clip-to-W quirk), cells
// left of column 0 only repeat pixels of the disk.
__global__ __launch_bounds__(256) void nms2d_push_kernel(const float* __restrict__ score, int H, int W,
                                                         const int* __restrict__ cells, int ncells, uint8_t* status,
                                                         const uint64_t* __restrict__ keys, const unsigned int* __restrict__ snap) {
    const unsigned int lo = snap[0], hi = snap[1];
    const int lane = threadIdx.x & 63;
    for (unsigned int i = lo + blockIdx.x * 4 + (threadIdx.x >> 6); i < hi; i += gridDim.x * 4) {
        const uint64_t kp = keys[i];
        const uint32_t p = (uint32_t)(kp & 0xffffffffull);
        const int py = (int)(p / (uint32_t)W), px = (int)(p % (uint32_t)W);
        for (int c = lane; c < ncells; c += 64) {
            const int code = cells[c];
            const int dy = code >> 16, dx = (code & 0xffff) - 32768;
            const int qy = py + dy, qx = px + dx;
            if ((unsigned)qy >= (unsigned)H || qx < 0) continue;
            uint32_t q;
            if (qx < W) q = (uint32_t)((size_t)qy * W + qx);
            else if (qy + 1 <= H - 1) q = (uint32_t)((size_t)qy * W + W);          // flat = qy*W + W == (qy+1, 0)
            else continue;
            if (status[q] == ST_UNDECIDED && prio_key(score[q], q) < kp) status[q] = ST_SUPPRESSED;
        }
    }
}
__global__ __launch_bounds__(256) void nms3d_push_kernel(const float* __restrict__ score, long long n,
                                                         const int* __restrict__ deltas, int ndelta, uint8_t* status,
                                                         const uint64_t* __restrict__ keys, const unsigned int* __restrict__ snap) {
    const unsigned int lo = snap[0], hi = snap[1];
    const int lane = threadIdx.x & 63;
    for (unsigned int i = lo + blockIdx.x * 4 + (threadIdx.x >> 6); i < hi; i += gridDim.x * 4) {
        const uint64_t kp = keys[i];
        const uint32_t p = (uint32_t)(kp & 0xffffffffull);
        for (int d = lane; d < ndelta; d += 64) {
            const long long q = (long long)p + deltas[d];
            if (q < 0 || q >= n) continue;
            if (status[q] == ST_UNDECIDED && prio_key(score[q], (uint32_t)q) < kp) status[q] = ST_SUPPRESSED;
        }
    }
}
// snap[1] = picks so far (the range the next push covers starts where the previous one ended)
__global__ void nms_snap_kernel(unsigned int* snap, const unsigned int* npicks) { snap[1] = *npicks; }

__global__ __launch_bounds__(256) void fill_u64_kernel(uint64_t* p, size_t lo, size_t hi, uint64_t v) {
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (size_t)gridDim.x * 256) p[i] = v;
}

// bitonic sort, descending, n a power of two.  Steps with partner distance j < 2048 run inside
// LDS (4096 keys per block); larger distances are one global compare-exchange sweep each.
__global__ __launch_bounds__(256) void bitonic_global_kernel(uint64_t* keys, size_t n, size_t j, size_t k) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n / 2; t += (size_t)gridDim.x * 256) {
        const size_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
        const size_t l = i | j;
        const bool desc = (i & k) == 0;
        const uint64_t a = keys[i], b = keys[l];
        if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[l] = a; }
    }
}

#define BITONIC_TILE 4096
__global__ __launch_bounds__(256) void bitonic_local_kernel(uint64_t* keys, size_t n, size_t k_lo, size_t k_hi,
                                                            size_t j_start) {
    // performs, on each 4096-key tile: for k = k_lo..k_hi (doubling): for j = (k == k_lo ? j_start : k/2) .. 1
    __shared__ uint64_t s[BITONIC_TILE];
    const size_t base = (size_t)blockIdx.x * BITONIC_TILE;
    for (int t = threadIdx.x; t < BITONIC_TILE; t += 256) s[t] = keys[base + t];
    __syncthreads();
    for (size_t k = k_lo; k <= k_hi; k <<= 1) {
        for (size_t j = (k == k_lo ? j_start : k >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < BITONIC_TILE / 2; t += 256) {
                const size_t i = (((size_t)t & ~(j - 1)) << 1) | ((size_t)t & (j - 1));
                const size_t l = i | j;
                const bool desc = ((base + i) & k) == 0;
                const uint64_t a = s[i], b = s[l];
                if (desc ? (a < b) : (a > b)) { s[i] = b; s[l] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < BITONIC_TILE; t += 256) keys[base + t] = s[t];
}

hipError_t bitonic_sort_desc(uint64_t* keys, size_t npow2, hipStream_t s) {
    if (npow2 < 2) return hipSuccess;
    if (npow2 < BITONIC_TILE) return hipErrorInvalidValue;   // caller pads to >= one tile
    const int tiles = (int)(npow2 / BITONIC_TILE);
    // all stages with k <= tile size: fully local
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(tiles), dim3(256), 0, s, keys, npow2, (size_t)2,
                       (size_t)BITONIC_TILE, (size_t)1);
    for (size_t k = (size_t)BITONIC_TILE * 2; k <= npow2; k <<= 1) {
        size_t j = k >> 1;
        for (; j >= BITONIC_TILE; j >>= 1) {
            const size_t half = npow2 / 2;
            int blocks = (int)((half + 255) / 256 < 8192 ? (half + 255) / 256 : 8192);
            hipLaunchKernelGGL(bitonic_global_kernel, dim3(blocks), dim3(256), 0, s, keys, npow2, j, k);
        }
        // remaining j = TILE/2 .. 1 of this k inside LDS
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(tiles), dim3(256), 0, s, keys, npow2, k, k, j);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nms_write_kernel(const uint64_t* __restrict__ keys, unsigned int n,
                                                        const float* __restrict__ score, int H, int W, int dims,
                                                        int32_t* __restrict__ coords, float* __restrict__ out_scores) {
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t p = (uint32_t)(keys[i] & 0xffffffffull);
        out_scores[i] = score[p];
        const uint32_t x = p % (uint32_t)W;
        const uint32_t t = p / (uint32_t)W;
        if (dims == 2) {
            coords[2 * i] = (int32_t)x;
            coords[2 * i + 1] = (int32_t)t;
        } else {
            coords[3 * i] = (int32_t)x;
            coords[3 * i + 1] = (int32_t)(t % (uint32_t)H);
            coords[3 * i + 2] = (int32_t)(t / (uint32_t)H);
        }
    }
}

// ---- host-side launch helpers (called from rt_nms.hip) -----------------------------------------
static inline int nblocks(size_t n, int cap = 8192) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < (size_t)cap ? b : (size_t)cap);
}

hipError_t nms_mark(const float* score, size_t n, float thr, uint8_t* status, uint32_t* cand, unsigned int* counters,
                    hipStream_t s) {
    hipLaunchKernelGGL(nms_mark_kernel, dim3(nblocks((n + 3) / 4, 4096)), dim3(256), 0, s, score, n, thr, status, cand,
                       counters);
    return hipGetLastError();
}
// One sweep = filter (thread per candidate, near cells) -> verify (wave per survivor, whole suppressor set; selects) ->
// snapshot of the pick count -> push (wave per new pick).  cnt[k]: length of list k (sweep k of a batch reads [k], appends
// its leftovers under [k + 1]); cnt_ver: length of this sweep's verify list; snap[k]: picks before sweep k.
hipError_t nms2d_sweep(const float* score, int H, int W, const int* near_cells, int n_near, const int* cells, int ncells,
                       uint8_t* status, const uint32_t* list_in, uint32_t* list_out, uint32_t* list_ver, unsigned int* cnt,
                       unsigned int* cnt_ver, unsigned int* snap, uint64_t* keys, unsigned int* npicks, size_t hint,
                       hipStream_t s) {
    hipLaunchKernelGGL(nms2d_filter_kernel, dim3(nblocks(hint, 16384)), dim3(256), 0, s, score, H, W, near_cells, n_near, status,
                       list_in, list_out, list_ver, cnt, cnt_ver);
    hipLaunchKernelGGL(nms2d_verify_kernel, dim3(nblocks(hint, 8192)), dim3(256), 0, s, score, H, W, cells, ncells, status,
                       list_ver, cnt_ver, keys, npicks);
    hipLaunchKernelGGL(nms_snap_kernel, dim3(1), dim3(1), 0, s, snap, npicks);
    hipLaunchKernelGGL(nms2d_push_kernel, dim3(nblocks(hint, 2048)), dim3(256), 0, s, score, H, W, cells, ncells, status,
                       keys, snap);
    return hipGetLastError();
}
hipError_t nms3d_sweep(const float* score, long long n, const int* near_deltas, int n_near, const int* deltas, int ndelta,
                       uint8_t* status, const uint32_t* list_in, uint32_t* list_out, uint32_t* list_ver, unsigned int* cnt,
                       unsigned int* cnt_ver, unsigned int* snap, uint64_t* keys, unsigned int* npicks, size_t hint,
                       hipStream_t s) {
    hipLaunchKernelGGL(nms3d_filter_kernel, dim3(nblocks(hint, 16384)), dim3(256), 0, s, score, n, near_deltas, n_near, status,
                       list_in, list_out, list_ver, cnt, cnt_ver);
    hipLaunchKernelGGL(nms3d_verify_kernel, dim3(nblocks(hint, 8192)), dim3(256), 0, s, score, n, deltas, ndelta, status,
                       list_ver, cnt_ver, keys, npicks);
    hipLaunchKernelGGL(nms_snap_kernel, dim3(1), dim3(1), 0, s, snap, npicks);
    hipLaunchKernelGGL(nms3d_push_kernel, dim3(nblocks(hint, 2048)), dim3(256), 0, s, score, n, deltas, ndelta, status,
                       keys, snap);
    return hipGetLastError();
}
hipError_t fill_u64(uint64_t* p, size_t lo, size_t hi, uint64_t v, hipStream_t s) {
    if (hi <= lo) return hipSuccess;
    hipLaunchKernelGGL(fill_u64_kernel, dim3(nblocks(hi - lo)), dim3(256), 0, s, p, lo, hi, v);
    return hipGetLastError();
}
hipError_t nms_write(const uint64_t* keys, unsigned int n, const float* score, int H, int W, int dims,
                     int32_t* coords, float* out_scores, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(nms_write_kernel, dim3(nblocks(n)), dim3(256), 0, s, keys, n, score, H, W, dims, coords,
                       out_scores);
    return hipGetLastError();
}

}  // namespace tpz
