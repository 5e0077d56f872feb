"""The mixture-fit oracle (oracle/gmm.py) against scikit-learn itself, the third-party dependency that
holds the algorithm the reference calls (data_transformer.py:38-49).  Same data, same initial
responsibilities (sklearn draws them from the RandomState it is handed)."""
import warnings

import numpy as np
import pytest

from oracle import gmm


def two_mode_column(n, seed):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.normal(0.0, 1.0, n // 2), rng.normal(10.0, 2.0, n - n // 2)])
    rng.shuffle(x)
    return x


@pytest.mark.parametrize('n,seed', [(200, 3), (64, 5)])
def test_oracle_matches_sklearn(n, seed):
    from sklearn.mixture import BayesianGaussianMixture
    x = two_mode_column(n, seed)
    K = 5
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        gm = BayesianGaussianMixture(n_components=K, covariance_type='full', weight_concentration_prior_type='dirichlet_process',
                                     weight_concentration_prior=1e-3, n_init=5, max_iter=2000, warm_start=True,
                                     init_params='random', random_state=np.random.RandomState(7))
        gm.fit(x.reshape(-1, 1))
    rs = np.random.RandomState(7)
    p, lb, n_iter, conv = gmm.fit(x, [gmm.random_resp(rs, n, K) for _ in range(5)])
    assert n_iter == gm.n_iter_ and conv == gm.converged_
    assert abs(lb - gm.lower_bound_) < 1e-9
    np.testing.assert_allclose(p['means'], gm.means_.ravel(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(p['cov'], gm.covariances_.ravel(), rtol=1e-12)
    np.testing.assert_allclose(p['pc'], gm.precisions_cholesky_.ravel(), rtol=1e-12)
    np.testing.assert_allclose(p['dof'], gm.degrees_of_freedom_, rtol=1e-12)
    np.testing.assert_allclose(p['mean_precision'], gm.mean_precision_, rtol=1e-12)
    np.testing.assert_allclose(p['a'], gm.weight_concentration_[0], rtol=1e-12)
    np.testing.assert_allclose(p['b'], gm.weight_concentration_[1], rtol=1e-12)
    np.testing.assert_allclose(gmm.weights(p), gm.weights_, rtol=1e-12)
    # and the responsibilities the encode uses (predict_proba)
    np.testing.assert_allclose(np.exp(gmm.e_step(x, p)), gm.predict_proba(x.reshape(-1, 1)), rtol=1e-9, atol=1e-300)
