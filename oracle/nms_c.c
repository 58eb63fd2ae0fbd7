/* Oracle (test infrastructure): plain-C restatement of the reference's greedy NMS.
 *
 * Follows topaz/algorithms.py:25-63 (2-D) and :66-103 (3-D) statement by statement: sort the
 * flat indices by descending score, walk them until the first score <= threshold, emit a pick
 * when the index is not in the suppressed set S, then add the pick's neighbourhood to S.
 * The python `set` S is a byte map here (2-D indices reach at most H*W + W, see the clip bound).
 *
 * Tie order: numpy's default argsort is not stable, so the order of equal scores in
 * `np.argsort(A)[::-1]` is implementation defined.  This restatement fixes it to
 * what a STABLE argsort reversed gives: equal scores are visited in DESCENDING flat index.
 * NaN compares like numpy's sort: greater than everything (visited first), and `A[i] <= thr`
 * is false for it.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/libnms_oracle.so oracle/nms_c.c   (oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const float* g_scores;

/* descending score, ties by descending index; NaN first (numpy sorts NaN to the end, reversed) */
static int cmp_desc(const void* pa, const void* pb) {
    const int64_t a = *(const int64_t*)pa, b = *(const int64_t*)pb;
    const float fa = g_scores[a], fb = g_scores[b];
    const int na = isnan(fa), nb = isnan(fb);
    if (na || nb) {
        if (na && nb) return a > b ? -1 : (a < b ? 1 : 0);
        return na ? -1 : 1;
    }
    if (fa > fb) return -1;
    if (fa < fb) return 1;
    return a > b ? -1 : (a < b ? 1 : 0);
}

/* returns the number of picks; coords[2*j] = x, coords[2*j+1] = y */
long nms2d_oracle(const float* x, long H, long W, long r, float threshold, float* scores, int32_t* coords) {
    const long n = H * W;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    /* flat = clip(y,0,H)*W + clip(x,0,W) <= H*W + W */
    uint8_t* S = (uint8_t*)calloc((size_t)(n + W + 1), 1);
    for (long i = 0; i < n; ++i) order[i] = i;
    g_scores = x;
    qsort(order, (size_t)n, sizeof(int64_t), cmp_desc);
    long j = 0;
    for (long t = 0; t < n; ++t) {
        const long i = order[t];
        if (x[i] <= threshold) break;                       /* algorithms.py:47-48 */
        if (!S[i]) {                                        /* :49 */
            const long xx = i % W, yy = i / W;              /* :51-52 */
            scores[j] = x[i];
            coords[2 * j] = (int32_t)xx;
            coords[2 * j + 1] = (int32_t)yy;
            ++j;
            for (long ii = -r; ii <= r; ++ii)               /* :28-32 mask, :58-61 */
                for (long jj = -r; jj <= r; ++jj) {
                    if (ii * ii + jj * jj > r * r) continue;
                    long yc = yy + ii, xc = xx + jj;
                    if (yc < 0) yc = 0;
                    if (yc > H) yc = H;                     /* np.clip(..., 0, x.shape[0]) */
                    if (xc < 0) xc = 0;
                    if (xc > W) xc = W;                     /* np.clip(..., 0, x.shape[1]) */
                    S[yc * W + xc] = 1;
                }
        }
    }
    free(order);
    free(S);
    return j;
}

/* r_scaled = scale*r as computed by the caller in double; coords[3*j..] = x, y, z */
long nms3d_oracle(const float* x, long D, long H, long W, double r_scaled, float threshold, float* scores,
                  int32_t* coords) {
    const long n = D * H * W;
    const long width = (long)ceil(r_scaled);
    const long zs = H * W, ys = W;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    uint8_t* S = (uint8_t*)calloc((size_t)n, 1);
    for (long i = 0; i < n; ++i) order[i] = i;
    g_scores = x;
    qsort(order, (size_t)n, sizeof(int64_t), cmp_desc);
    long j = 0;
    for (long t = 0; t < n; ++t) {
        const long i = order[t];
        if (x[i] <= threshold) break;
        if (!S[i]) {
            const long zz = i / zs, yy = (i % zs) / ys, xx = i % ys;   /* np.unravel_index */
            scores[j] = x[i];
            coords[3 * j] = (int32_t)xx;
            coords[3 * j + 1] = (int32_t)yy;
            coords[3 * j + 2] = (int32_t)zz;
            ++j;
            for (long ii = -width; ii <= width; ++ii)
                for (long jj = -width; jj <= width; ++jj)
                    for (long kk = -width; kk <= width; ++kk) {
                        if ((double)(ii * ii + jj * jj + kk * kk) > r_scaled * r_scaled) continue;
                        const long q = i + ii * zs + jj * ys + kk;     /* S.add(i + delta): no clipping */
                        if (q >= 0 && q < n) S[q] = 1;
                    }
        }
    }
    free(order);
    free(S);
    return j;
}
