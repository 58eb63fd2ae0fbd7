// Statistics and elementwise entry points: mean / std, GMM fit (topaz normalize), affine, normalise.
#include "rt_internal.h"

extern "C" {
int tpz_mean_std(tpz_ctx* ctx, const float* d_x, size_t n, int unbiased, float* h_mean_std) {
    if (!ctx || !d_x || !h_mean_std || n == 0) return fail(ctx, "tpz_mean_std: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    float* out = next_nrm(ctx);
    // present the vector as rows of <= 2^20 elements so int geometry cannot overflow
    const int Wv = (int)std::min<size_t>(n, (size_t)1 << 20);
    const size_t rows = n / Wv;
    if (rows * (size_t)Wv != n) {
        // fall back to a single row when n is not a multiple (n < 2^31 required)
        if (n >= ((size_t)1 << 31)) return fail(ctx, "tpz_mean_std: n too large for a ragged vector");
        HIPCHK(ctx, launch_meanstd(d_x, 1, 1, (int)n, 0, (int)n, unbiased, 0, nullptr, ctx->d_part, PART_BLOCKS, out, ctx->stream));
    } else {
        HIPCHK(ctx, launch_meanstd(d_x, 1, (int)rows, Wv, 0, Wv, unbiased, 0, nullptr, ctx->d_part, PART_BLOCKS, out, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(h_mean_std, out, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- 2-component Gaussian mixture fit (topaz normalize) ---------------------------------------------------
// topaz/stats.py:87-117 norm_fit + :120-203 gmm_fit (share_var = True), evaluated in fp64 from the sufficient
// statistics of gmm_pass_kernel: one device pass per EM iteration, the scalar M-step on the host.
static double beta_logpdf(double x, double a, double b) {
    // scipy.stats.beta.logpdf: xlog1py(b-1, -x) + xlogy(a-1, x) - betaln(a, b)   (0 * log(0) = 0)
    const double t1 = (b - 1.0) == 0.0 ? 0.0 : (b - 1.0) * std::log1p(-x);
    const double t2 = (a - 1.0) == 0.0 ? 0.0 : (a - 1.0) * std::log(x);
    return t1 + t2 - (std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b));
}

int tpz_gmm_fit(tpz_ctx* ctx, const float* d_x, size_t n, const double* pis, const double* splits, int n_init,
                double alpha, double beta, double scale, int num_iters, double tol, double* mus, double* stds,
                double* pis_out, double* logps) {
    if (!ctx || !d_x || !pis || !splits || !mus || !stds || !pis_out || !logps || n < 2 || n_init < 1)
        return fail(ctx, "tpz_gmm_fit: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double *d_par = nullptr, *d_out = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&d_par, 8 * sizeof(double)));
    if (hipMalloc((void**)&d_out, 8 * sizeof(double)) != hipSuccess) { (void)hipFree(d_par); return fail(ctx, "hipMalloc failed"); }
    int rc = 0;
    const double N = (double)n;
    auto pass = [&](int mode, const double (&par)[6], double (&S)[7]) -> int {
        if (hipMemcpyAsync(d_par, par, 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return 1;
        prof_begin(ctx, 2, 0);
        hipError_t e = launch_gmm_pass(d_x, n, mode, d_par, ctx->d_part, 256, d_out, ctx->stream);
        prof_end(ctx);
        if (e != hipSuccess) return 1;
        if (hipMemcpyAsync(S, d_out, 7 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return 1;
        return hipStreamSynchronize(ctx->stream) != hipSuccess;
    };
    // global moments: hard pass with split = +inf puts everything in component 0
    double S[7];
    {
        const double par[6] = {INFINITY, 0, 0, 0, 0, 0};
        if (pass(0, par, S)) rc = fail(ctx, "tpz_gmm_fit: device pass failed");
    }
    const double mu_all = S[3] / N;
    const double var_unbiased = (S[5] - 2.0 * mu_all * S[3] + mu_all * mu_all * N) / (N - 1.0);   // torch .var()
    for (int i = 0; i < n_init && !rc; ++i) {
        double pi = pis[i];
        if (pi == 1.0) {
            // single-component model (stats.py:100-103): note the reference adds beta.PDF(1), not its logarithm
            const double pdf1 = beta == 1.0 ? std::exp(-(std::lgamma(alpha) + std::lgamma(beta) - std::lgamma(alpha + beta)))
                                            : (beta > 1.0 ? 0.0 : INFINITY);
            logps[i] = scale * (-(N - 1.0) / 2.0 - N * 0.5 * std::log(2.0 * M_PI * var_unbiased)) + pdf1;
            mus[i] = mu_all;
            stds[i] = std::sqrt(var_unbiased);
            pis_out[i] = 1.0;
            continue;
        }
        auto m_step = [&](const double (&T)[7], double& mu0, double& mu1, double& var) {
            mu0 = T[1] > 0 ? T[3] / T[1] : mu_all;
            mu1 = T[2] > 0 ? T[4] / T[2] : mu_all;
            var = ((T[5] - 2.0 * mu0 * T[3] + mu0 * mu0 * T[1]) + (T[6] - 2.0 * mu1 * T[4] + mu1 * mu1 * T[2])) / N;
        };
        double mu0, mu1, var;
        {
            const double par[6] = {splits[i], 0, 0, 0, 0, 0};
            if (pass(0, par, S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
        }
        m_step(S, mu0, mu1, var);
        auto e_step = [&](double (&T)[7]) -> int {
            const double par[6] = {mu0, mu1, var, var, std::log1p(-pi), std::log(pi)};
            return pass(1, par, T);
        };
        if (e_step(S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
        double logp = scale * S[0] + beta_logpdf(pi, alpha, beta);
        double logp_cur = logp;
        for (int it = 1; it <= num_iters; ++it) {
            // M-step from the assignments of the last E-step (S), MAP estimate of pi under the Beta prior
            const double a_ = alpha + S[2], b_ = beta + N - S[2];
            pi = (a_ - 1.0) / (a_ + b_ - 2.0);
            m_step(S, mu0, mu1, var);
            if (e_step(S)) { rc = fail(ctx, "tpz_gmm_fit: device pass failed"); break; }
            logp = scale * S[0] + beta_logpdf(pi, alpha, beta);
            if (logp - logp_cur <= tol) break;
            logp_cur = logp;
        }
        logps[i] = logp;
        mus[i] = mu1;
        stds[i] = std::sqrt(var);
        pis_out[i] = pi;
    }
    (void)hipFree(d_par);
    (void)hipFree(d_out);
    return rc;
}

int tpz_affine(tpz_ctx* ctx, const float* d_x, size_t n, float scale, float shift, float* d_y) {
    if (!ctx || !d_x || !d_y) return fail(ctx, "tpz_affine: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prof_begin(ctx, 2, 0);
    hipError_t e = launch_affine(d_x, d_y, n, scale, shift, ctx->stream);
    prof_end(ctx);
    HIPCHK(ctx, e);
    return 0;
}

int tpz_normalize(tpz_ctx* ctx, const float* d_x, size_t n, float mean, float std, float* d_y) {
    if (!ctx || !d_x || !d_y) return fail(ctx, "tpz_normalize: bad arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    prof_begin(ctx, 2, 0);
    hipError_t e = launch_normalize(d_x, d_y, n, mean, std, ctx->stream);
    prof_end(ctx);
    HIPCHK(ctx, e);
    return 0;
}

}  // extern "C"
