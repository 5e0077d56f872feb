"""Variational Bayesian Gaussian mixtures for the continuous columns, fitted on the device.

Host side of csrc/gmm.hip.  Replaces the ``BayesianGaussianMixture(...).fit(column)`` call of the
reference (/root/reference/src/dp_cgans/synthesizers/data_transformer.py:36-57; scikit-learn's
algorithm, restated for the tests in oracle/gmm.py): all columns and all ``n_init`` initialisations
run side by side as independent fits, two launches per iteration, until each fit's lower bound moves
by less than ``tol`` (or ``max_iter``); per column the initialisation with the best lower bound wins.
"""
import numpy as np
import torch

from . import ops


class MixtureFit(object):
    """Parameters of one fitted column, named as sklearn names them (``means_`` -> ``means`` ...)."""

    def __init__(self, table, lower_bound, n_iter, converged):
        (self.means, self.precisions_cholesky, _, self.covariances, a, b, self.mean_precision,
         self.degrees_of_freedom) = table
        self.weight_concentration = (a, b)
        s = a + b                                           # _bayesian_mixture.py:860-871
        w = a / s * np.hstack((1.0, np.cumprod((b / s)[:-1])))
        self.weights = w / w.sum()
        self.lower_bound, self.n_iter, self.converged = float(lower_bound), int(n_iter), bool(converged)


class DeviceBayesianGMM(object):
    def __init__(self, n_components=5, weight_concentration_prior=1e-3, n_init=5, max_iter=2000, tol=1e-3,
                 reg_covar=1e-6, device='cuda:0', seed=0, poll=32):
        if not 1 <= n_components <= 16:
            raise ValueError('1 <= n_components <= 16')
        self.K, self.wcp, self.n_init, self.max_iter = int(n_components), float(weight_concentration_prior), int(n_init), int(max_iter)
        self.tol, self.reg_covar, self.device, self.seed, self.poll = float(tol), float(reg_covar), torch.device(device), seed, poll
        self.iterations_run = 0

    def fit_columns(self, columns, resp_inj=None):
        """columns: fp64 [ncol, N] (numpy or device tensor).  resp_inj: optional fp64
        [ncol, n_init, N, K] initial responsibilities (rows sum to one) in place of Philox uniforms.
        -> list of MixtureFit, one per column."""
        dev, K, n_init = self.device, self.K, self.n_init
        x = torch.as_tensor(columns, dtype=torch.float64).to(dev).contiguous()
        if x.dim() != 2:
            raise ValueError('columns must be [ncol, N]')
        ncol, N = x.shape
        if ncol == 0:
            return []
        if N < 2 or N < K:
            raise ValueError('Expected n_samples >= n_components but got n_components = %d, n_samples = %d' % (K, N))
        if not bool(torch.isfinite(x).all()):
            raise ValueError('Input X contains NaN or infinity.')
        F = ncol * n_init
        if F > 65535:
            raise ValueError('too many fits in one call: %d' % F)
        fit_col = torch.arange(ncol, dtype=torch.int32, device=dev).repeat_interleave(n_init)     # f = col * n_init + init
        prior = torch.stack([x.mean(dim=1), x.var(dim=1, unbiased=True)], dim=1).repeat_interleave(n_init, dim=0).contiguous()
        R = int(max(1, min((N + 511) // 512, 128)))
        params = torch.zeros(F, 8, K, dtype=torch.float64, device=dev)
        info = torch.zeros(F, 4, dtype=torch.float64, device=dev)
        part = torch.zeros(F, R, 3 * K + 1, dtype=torch.float64, device=dev)
        rng = ops.new_rng(self.seed ^ 0x6d6d, dev)
        if resp_inj is not None:
            resp_inj = torch.as_tensor(resp_inj, dtype=torch.float64).to(dev).reshape(F, N, K).contiguous()
        hyper = dict(weight_concentration_prior=self.wcp, reg_covar=self.reg_covar, tol=self.tol, max_iter=self.max_iter)
        ops.gmm_estep(x, fit_col, params, prior, info, part, True, resp_inj=resp_inj, rng=rng)
        ops.gmm_mstep(part, prior, params, info, True, **hyper)
        it = 0
        while it < self.max_iter:
            for _ in range(min(self.poll, self.max_iter - it)):
                ops.gmm_estep(x, fit_col, params, prior, info, part, False)
                ops.gmm_mstep(part, prior, params, info, False, **hyper)
                it += 1
            if bool((info[:, 3] != 0).all()):
                break
        self.iterations_run = it
        P = params.cpu().numpy().reshape(ncol, n_init, 8, K)
        I = info.cpu().numpy().reshape(ncol, n_init, 4)
        out = []
        for c in range(ncol):
            best = int(np.argmax(I[c, :, 0]))              # first of equal bounds, as `>` in _base.py:264
            out.append(MixtureFit(P[c, best], I[c, best, 0], I[c, best, 1], I[c, best, 2]))
        return out
