#!/usr/bin/env python3
"""Wall time of the continuous-column mixture fit (DataTransformer.fit) on one MI355X:
the 20 continuous columns of the T1M schema (col j ~ mixture of 1 + j%3 Gaussians), n_init = 5,
K = 5, i.e. 100 fits of 1M rows side by side.  --sklearn N also times scikit-learn on ONE column of
N rows on the host for scale."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def raw_columns(n_cont, N, seed=1234):
    rng = np.random.default_rng(seed)
    cols = []
    for j in range(n_cont):
        m = 1 + j % 3
        comp = rng.integers(m, size=N)
        cols.append(rng.normal(10.0 * comp, 1.0 + 0.25 * comp))
    return np.stack(cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1000000)
    ap.add_argument('--cols', type=int, default=20)
    ap.add_argument('--max-iter', type=int, default=2000)
    ap.add_argument('--sklearn', type=int, default=0)
    a = ap.parse_args()
    import torch
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    x = raw_columns(a.cols, a.rows)
    xd = torch.from_numpy(x).to('cuda:0')
    m = DeviceBayesianGMM(device='cuda:0', seed=1, max_iter=a.max_iter)
    m.fit_columns(xd[:1, :10000])                     # load the library, warm up
    torch.cuda.synchronize()
    t = time.perf_counter()
    fits = m.fit_columns(xd)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print('device: %d columns x %d rows x 5 inits: %.3f s, %d iterations queued (%.1f us per iteration of all fits)'
          % (a.cols, a.rows, dt, m.iterations_run, 1e6 * dt / max(m.iterations_run, 1)), flush=True)
    for j, f in enumerate(fits):
        print('  col %2d (%d modes): n_iter %4d converged %d lower bound %.3f kept %d weights %s' % (
            j, 1 + j % 3, f.n_iter, f.converged, f.lower_bound, int((f.weights > 1e-3).sum()), np.round(f.weights, 4)), flush=True)
    if a.sklearn:
        import warnings
        from sklearn.mixture import BayesianGaussianMixture
        xs = x[2, :a.sklearn].reshape(-1, 1)
        gm = BayesianGaussianMixture(n_components=5, covariance_type='full', weight_concentration_prior_type='dirichlet_process',
                                     weight_concentration_prior=1e-3, n_init=5, max_iter=a.max_iter, warm_start=True,
                                     init_params='random', random_state=np.random.RandomState(0))
        t = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            gm.fit(xs)
        dt = time.perf_counter() - t
        print('sklearn (host): 1 column x %d rows x 5 inits: %.1f s, best n_iter %d converged %s' % (a.sklearn, dt, gm.n_iter_, gm.converged_), flush=True)


if __name__ == '__main__':
    main()
