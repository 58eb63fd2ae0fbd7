"""Oracle (test infrastructure): float64 NumPy restatement of the pixel-mixture fit behind `topaz normalize`.

Follows topaz/stats.py: gmm_fit (:120-203, share_var=True), norm_fit (:87-117), normalize (:37-84).  The
reference evaluates the fit in float32 torch and stops on `logp - logp_cur <= 1e-3`, a test its float32
log-likelihood (magnitude 1e5..1e7) can only resolve to ~0.1, so two correct evaluations may stop an iteration
apart.  Pinned against the reference itself (tests/golden/normalize_*.npz from oracle/make_golden.py): all twelve
fits agree to 3e-6 relative in mu / std, 1e-4 in pi, 2e-6 relative in the log-likelihood, the same model is
selected and the normalised image agrees to 5e-7."""
from __future__ import annotations

import math

import numpy as np

INIT_PIS = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.98, 1.0)


def beta_logpdf(x, a, b):
    t1 = 0.0 if b == 1 else (b - 1) * math.log1p(-x)
    t2 = 0.0 if a == 1 else (a - 1) * math.log(x)
    return t1 + t2 - (math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b))


def gmm_fit(x, pi, split, alpha, beta, scale=1.0, tol=1e-3, num_iters=100):
    """stats.py:120-203.  Returns (logp, mu1, var, pi)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    n = x.size
    mu = x.mean()
    p0 = (x <= split).astype(np.float64)
    p1 = 1 - p0

    def m_step(p0, p1):
        s0, s1 = p0.sum(), p1.sum()
        mu0 = (x * p0).sum() / s0 if s0 > 0 else mu
        mu1 = (x * p1).sum() / s1 if s1 > 0 else mu
        var = np.mean(p0 * (x - mu0) ** 2 + p1 * (x - mu1) ** 2)
        return mu0, mu1, var

    def e_step(mu0, mu1, var, pi):
        l0 = -(x - mu0) ** 2 / 2 / var - 0.5 * math.log(2 * math.pi * var) + math.log1p(-pi)
        l1 = -(x - mu1) ** 2 / 2 / var - 0.5 * math.log(2 * math.pi * var) + math.log(pi)
        ma = np.maximum(l0, l1)
        Z = ma + np.log(np.exp(l0 - ma) + np.exp(l1 - ma))
        return l0, l1, Z

    mu0, mu1, var = m_step(p0, p1)
    l0, l1, Z = e_step(mu0, mu1, var, pi)
    logp = scale * Z.sum() + beta_logpdf(pi, alpha, beta)
    logp_cur = logp
    for _ in range(1, num_iters + 1):
        p0, p1 = np.exp(l0 - Z), np.exp(l1 - Z)
        s = p1.sum()
        a, b = alpha + s, beta + n - s
        pi = (a - 1) / (a + b - 2)
        mu0, mu1, var = m_step(p0, p1)
        l0, l1, Z = e_step(mu0, mu1, var, pi)
        logp = scale * Z.sum() + beta_logpdf(pi, alpha, beta)
        if logp - logp_cur <= tol:
            break
        logp_cur = logp
    return logp, mu1, var, pi


def norm_fit(x, alpha=900, beta=1, scale=1.0, num_iters=100, tol=1e-3):
    """stats.py:87-117.  Returns (mus, stds, pis, logps) of the twelve initialisations."""
    x32 = np.asarray(x, dtype=np.float32).ravel()
    pis = np.array(INIT_PIS, dtype=np.float64)
    splits = np.quantile(x32, 1 - pis)
    x = x32.astype(np.float64)
    n = x.size
    mus, stds, logps = np.zeros(len(pis)), np.zeros(len(pis)), np.zeros(len(pis))
    for i, (pi, split) in enumerate(zip(pis.copy(), splits)):
        if pi == 1:
            mu, var = x.mean(), x.var(ddof=1)
            # the reference adds beta.PDF(1; alpha, beta) here, not its logarithm (stats.py:103)
            pdf1 = math.exp(-(math.lgamma(alpha) + math.lgamma(beta) - math.lgamma(alpha + beta))) if beta == 1 else (0.0 if beta > 1 else math.inf)
            logps[i] = scale * np.sum(-(x - mu) ** 2 / 2 / var - 0.5 * math.log(2 * math.pi * var)) + pdf1
            mus[i], stds[i] = mu, math.sqrt(var)
        else:
            logp, mu1, var, pi_fit = gmm_fit(x, pi, split, alpha, beta, scale=scale, tol=tol, num_iters=num_iters)
            logps[i], mus[i], stds[i], pis[i] = logp, mu1, math.sqrt(var), pi_fit
    return mus, stds, pis, logps
