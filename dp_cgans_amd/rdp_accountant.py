"""RDP accounting of the sampled Gaussian mechanism (host scalar code, called once per epoch).

Same call surface as the reference's vendored TF-Privacy module
(/root/reference/src/dp_cgans/functions/rdp_accountant.py:341-360 compute_rdp, :594-625
get_privacy_spent) restated from the published algorithm (Mironov, Talwar, Zhang 2019, "Renyi
Differential Privacy of the Sampled Gaussian Mechanism"; conversion bound of Canonne, Kamath,
Steinke 2020, Prop. 12).  Checked against tests/golden/rdp.npz captured from the reference.
Out of kernel scope: a few hundred scalar operations per epoch.
"""
import math

import numpy as np
from scipy import special


def _logaddexp(a, b):
    lo, hi = (a, b) if a < b else (b, a)
    if lo == -np.inf:
        return hi
    return math.log1p(math.exp(lo - hi)) + hi


def _logsubexp(a, b):
    if a < b:
        raise ValueError('The result of subtraction must be non-negative.')
    if b == -np.inf:
        return a
    if a == b:
        return -np.inf
    try:
        return math.log(math.expm1(a - b)) + b
    except OverflowError:
        return a


def _log_erfc(x):
    return math.log(2) + special.log_ndtr(-x * 2 ** .5)


def _log_a_integer(q, sigma, alpha):
    acc = -np.inf
    for i in range(alpha + 1):
        log_binom = special.gammaln(alpha + 1) - special.gammaln(i + 1) - special.gammaln(alpha - i + 1)
        term = log_binom + i * math.log(q) + (alpha - i) * math.log(1 - q) + (i * i - i) / (2 * sigma ** 2)
        acc = _logaddexp(acc, term)
    return float(acc)


def _log_a_fractional(q, sigma, alpha):
    lo, hi = -np.inf, -np.inf
    z0 = sigma ** 2 * math.log(1 / q - 1) + .5
    i = 0
    while True:
        coef = special.binom(alpha, i)
        log_coef = math.log(abs(coef))
        j = alpha - i
        t0 = log_coef + i * math.log(q) + j * math.log(1 - q)
        t1 = log_coef + j * math.log(q) + i * math.log(1 - q)
        e0 = math.log(.5) + _log_erfc((i - z0) / (math.sqrt(2) * sigma))
        e1 = math.log(.5) + _log_erfc((z0 - j) / (math.sqrt(2) * sigma))
        s0 = t0 + (i * i - i) / (2 * sigma ** 2) + e0
        s1 = t1 + (j * j - j) / (2 * sigma ** 2) + e1
        if coef > 0:
            lo, hi = _logaddexp(lo, s0), _logaddexp(hi, s1)
        else:
            lo, hi = _logsubexp(lo, s0), _logsubexp(hi, s1)
        i += 1
        if max(s0, s1) < -30:
            break
    return _logaddexp(lo, hi)


def _rdp_one_order(q, sigma, alpha):
    if q == 0:
        return 0
    if q == 1.:
        return alpha / (2 * sigma ** 2)
    if np.isinf(alpha):
        return np.inf
    if float(alpha).is_integer():
        log_a = _log_a_integer(q, sigma, int(alpha))
    else:
        log_a = _log_a_fractional(q, sigma, alpha)
    return log_a / (alpha - 1)


def compute_rdp(q, noise_multiplier, steps, orders):
    """RDP of `steps` compositions of the sampled Gaussian mechanism at each order."""
    if np.isscalar(orders):
        rdp = _rdp_one_order(q, noise_multiplier, orders)
    else:
        rdp = np.array([_rdp_one_order(q, noise_multiplier, a) for a in orders])
    return rdp * steps


def _eps_from_rdp(orders, rdp, delta):
    orders, rdp = np.atleast_1d(orders), np.atleast_1d(rdp)
    if delta <= 0:
        raise ValueError('Privacy failure probability bound delta must be >0.')
    if len(orders) != len(rdp):
        raise ValueError('Input lists must have the same length.')
    eps = []
    for a, r in zip(orders, rdp):
        if a < 1:
            raise ValueError('Renyi divergence order must be >=1.')
        if r < 0:
            raise ValueError('Renyi divergence must be >=0.')
        if delta ** 2 + math.expm1(-r) >= 0:
            e = 0
        elif a > 1.01:
            e = r + math.log1p(-1 / a) - math.log(delta * a) / (a - 1)
        else:
            e = np.inf
        eps.append(e)
    i = int(np.argmin(eps))
    return max(0, eps[i]), orders[i]


def _delta_from_rdp(orders, rdp, eps):
    orders, rdp = np.atleast_1d(orders), np.atleast_1d(rdp)
    if eps < 0:
        raise ValueError('Value of privacy loss bound epsilon must be >=0.')
    if len(orders) != len(rdp):
        raise ValueError('Input lists must have the same length.')
    logd = []
    for a, r in zip(orders, rdp):
        if a < 1:
            raise ValueError('Renyi divergence order must be >=1.')
        if r < 0:
            raise ValueError('Renyi divergence must be >=0.')
        v = 0.5 * math.log1p(-math.exp(-r))
        if a > 1.01:
            v = min(v, (a - 1) * (r - eps + math.log1p(-1 / a)) - math.log(a))
        logd.append(v)
    i = int(np.argmin(logd))
    return min(math.exp(logd[i]), 1.), orders[i]


def get_privacy_spent(orders, rdp, target_eps=None, target_delta=None):
    """-> (eps, delta, optimal order); exactly one of target_eps / target_delta must be given."""
    if target_eps is None and target_delta is None:
        raise ValueError('Exactly one out of eps and delta must be None. (Both are).')
    if target_eps is not None and target_delta is not None:
        raise ValueError('Exactly one out of eps and delta must be None. (None is).')
    if target_eps is not None:
        delta, order = _delta_from_rdp(orders, rdp, target_eps)
        return target_eps, delta, order
    eps, order = _eps_from_rdp(orders, rdp, target_delta)
    return eps, target_delta, order
