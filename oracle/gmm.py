"""TEST INFRASTRUCTURE ONLY (oracle) - never imported by the product path.

CPU restatement of the variational Bayesian Gaussian mixture the reference fits to every continuous
column (/root/reference/src/dp_cgans/synthesizers/data_transformer.py:36-57):

    BayesianGaussianMixture(n_components=5, covariance_type='full',
                            weight_concentration_prior_type='dirichlet_process',
                            weight_concentration_prior=1e-3, n_init=5, max_iter=2000,
                            warm_start=True, init_params='random')          # tol 1e-3, reg_covar 1e-6

The algorithm lives in a third-party dependency, scikit-learn (requirements.txt:9 pins >=1.3.2; the
image holds 1.7.2): sklearn/mixture/_base.py (fit_predict, random initial responsibilities) and
_bayesian_mixture.py (Bishop PRML 10.2 with normalised Wishart scale matrices, stick-breaking weights,
a lower bound without its constant terms and NOT divided by the number of rows).  Restated here for
one feature; tests/test_oracle_golden.py pins it to sklearn itself run on the same initial
responsibilities (sklearn is importable here and on the GPU box).

All quantities fp64.  State of one fit with K components:
    nk, means, mean_precision, dof, cov (already divided by dof), pc = cov**-0.5, (a, b) of the
    stick-breaking Beta posteriors.
"""
import numpy as np
from scipy.special import betaln, digamma, gammaln

TOL = 1e-3
REG_COVAR = 1e-6
MAX_ITER = 2000
EPS10 = 10 * np.finfo(np.float64).eps


def priors(x):
    """mean_prior = mean(x), covariance_prior = cov(x) (ddof 1), mean_precision_prior = 1,
    degrees_of_freedom_prior = n_features = 1 (_bayesian_mixture.py:432-489)."""
    x = np.asarray(x, dtype=np.float64)
    return float(x.mean()), float(np.cov(x)), 1.0, 1.0


def m_step(x, resp, prior, wcp=1e-3):
    """_initialize / _m_step (:515-530,725-743) from responsibilities [N, K]."""
    mean_prior, cov_prior, mpp, dofp = prior
    nk = resp.sum(axis=0) + EPS10
    xk = resp.T @ x / nk
    diff = x[None, :] - xk[:, None]
    sk = (resp.T * diff * diff).sum(axis=1) / nk + REG_COVAR
    a = 1.0 + nk
    b = wcp + np.hstack((np.cumsum(nk[::-1])[-2::-1], 0))
    mean_precision = mpp + nk
    means = (mpp * mean_prior + nk * xk) / mean_precision
    dof = dofp + nk
    cov = (cov_prior + nk * sk + nk * mpp / mean_precision * (xk - mean_prior) ** 2) / dof
    pc = 1.0 / np.sqrt(cov)
    return dict(a=a, b=b, mean_precision=mean_precision, means=means, dof=dof, cov=cov, pc=pc)


def log_weights(p):
    """_estimate_log_weights (:745-756)."""
    ds = digamma(p['a'] + p['b'])
    return digamma(p['a']) - ds + np.hstack((0, np.cumsum(digamma(p['b']) - ds)[:-1]))


def weighted_log_prob(x, p):
    """_estimate_log_prob + log weights (:763-779) for one feature: A_k - .5 ((x - mu_k) pc_k)^2."""
    A = (log_weights(p) + np.log(p['pc']) - 0.5 * np.log(2.0 * np.pi) - 0.5 * np.log(p['dof'])
         + 0.5 * (np.log(2.0) + digamma(0.5 * p['dof']) - 1.0 / p['mean_precision']))
    y = (x[:, None] - p['means'][None, :]) * p['pc'][None, :]
    return A[None, :] - 0.5 * y * y


def e_step(x, p):
    wlp = weighted_log_prob(x, p)
    mx = wlp.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(wlp - mx).sum(axis=1))
    return wlp - lse[:, None]


def lower_bound(log_resp, p):
    """_compute_lower_bound (:781-838) with n_features = 1."""
    ldpc = np.log(p['pc']) - 0.5 * np.log(p['dof'])
    log_wishart = np.sum(-(p['dof'] * ldpc + p['dof'] * 0.5 * np.log(2.0) + gammaln(0.5 * p['dof'])))
    log_norm_weight = -np.sum(betaln(p['a'], p['b']))
    return (-np.sum(np.exp(log_resp) * log_resp) - log_wishart - log_norm_weight
            - 0.5 * np.sum(np.log(p['mean_precision'])))


def weights(p):
    """_set_parameters (:860-871): expected stick-breaking weights, normalised."""
    s = p['a'] + p['b']
    w = p['a'] / s * np.hstack((1, np.cumprod((p['b'] / s)[:-1])))
    return w / w.sum()


def fit_one(x, resp0, max_iter=MAX_ITER, tol=TOL):
    """One initialisation (_base.py:236-260): returns (params, lower_bound, n_iter, converged)."""
    x = np.asarray(x, dtype=np.float64)
    prior = priors(x)
    p = m_step(x, np.asarray(resp0, dtype=np.float64), prior)
    lb = -np.inf
    converged = False
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        prev = lb
        log_resp = e_step(x, p)
        p = m_step(x, np.exp(log_resp), prior)
        lb = lower_bound(log_resp, p)
        if abs(lb - prev) < tol:
            converged = True
            break
    return p, lb, n_iter, converged


def fit(x, resp0_list, max_iter=MAX_ITER, tol=TOL):
    """n_init initialisations, best lower bound wins, the first on ties (_base.py:233-269)."""
    best = None
    for resp0 in resp0_list:
        p, lb, n_iter, conv = fit_one(x, resp0, max_iter, tol)
        if best is None or lb > best[1]:
            best = (p, lb, n_iter, conv)
    return best


def random_resp(random_state, n, K):
    """init_params='random' (_base.py:121-125)."""
    resp = random_state.uniform(size=(n, K))
    return resp / resp.sum(axis=1)[:, None]
