"""Oracle (TEST INFRASTRUCTURE ONLY): mode-specific encode of DataTransformer, numpy fp64.

Reference: /root/reference/src/dp_cgans/synthesizers/data_transformer.py
  _transform_continuous :102-126, transform :133-150, _inverse_transform_continuous :152-175.
Third-party arithmetic restated: sklearn.mixture.BayesianGaussianMixture.predict_proba
(scikit-learn >=1.3.2 pinned by the reference's requirements.txt:9; 1.7.2 installed when the
fixtures were captured) for one feature, 'full' covariance, dirichlet_process weights:
  log w_k   = psi(a_k) - psi(a_k+b_k) + sum_{j<k} (psi(b_j) - psi(a_j+b_j))
  log p_k(x)= log pc_k - .5 log(2 pi) - .5 ((x-mu_k) pc_k)^2 - .5 log dof_k
              + .5 (log 2 + psi(dof_k/2) - 1/mean_precision_k)
  resp      = softmax_k(log w_k + log p_k(x))
np.random.choice(arange(n), p=p) is restated as searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right')
with the uniform u injected.
"""
import numpy as np
from scipy.special import digamma, logsumexp


def gm_constants(wc_a, wc_b, mean_precision, dof, prec_chol):
    """A_k: everything in the weighted log-probability that does not depend on x."""
    dsum = digamma(wc_a + wc_b)
    da, db = digamma(wc_a), digamma(wc_b)
    logw = da - dsum + np.hstack((0.0, np.cumsum(db - dsum)[:-1]))
    return (logw + np.log(prec_chol) - 0.5 * np.log(2 * np.pi) - 0.5 * np.log(dof)
            + 0.5 * (np.log(2.0) + digamma(0.5 * dof) - 1.0 / mean_precision))


def predict_proba(x, means, prec_chol, A):
    y = (np.asarray(x, dtype=np.float64).reshape(-1, 1) - means.reshape(1, -1)) * prec_chol.reshape(1, -1)
    wlp = A.reshape(1, -1) - 0.5 * (y * y)
    return np.exp(wlp - logsumexp(wlp, axis=1, keepdims=True))


def transform_continuous(x, means, covariances, valid, proba, u):
    """-> (value [N], component [N] (index among valid components), onehot [N, n_valid])."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    stds = np.sqrt(covariances).reshape(1, -1)
    norm = ((x - means.reshape(1, -1)) / (4 * stds))[:, valid]
    p = proba[:, valid]
    N, nv = p.shape
    comp = np.zeros(N, dtype=np.int64)
    for i in range(N):
        pt = p[i] + 1e-6
        pt = pt / pt.sum()
        cdf = pt.cumsum()
        cdf /= cdf[-1]
        comp[i] = cdf.searchsorted(u[i], side='right')
    val = np.clip(norm[np.arange(N), comp], -.99, .99)
    onehot = np.zeros((N, nv), dtype=np.float64)
    onehot[np.arange(N), comp] = 1
    return val, comp, onehot


def inverse_continuous(value, comp_probs, means, covariances, valid):
    """argmax component, de-normalise (no sigmas)."""
    v = np.clip(value, -1, 1)
    full = np.ones((len(value), len(means))) * -100
    full[:, valid] = comp_probs
    k = np.argmax(full, axis=1)
    return v * 4 * np.sqrt(covariances)[k] + means[k]
