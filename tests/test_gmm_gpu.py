"""Device mixture fit (csrc/gmm.hip through dpc_gmm_estep / dpc_gmm_mstep) against oracle/gmm.py.
fp64 throughout; the sums run in a different order than numpy's, so parameters are compared to 1e-8
relative and the lower bound to 1e-7 absolute (its convergence tolerance is 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import gmm

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def columns(n):
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.normal(0.0, 1.0, n // 2), rng.normal(10.0, 2.0, n - n // 2)])
    b = rng.normal(-3.0, 0.5, n)
    c = np.concatenate([np.zeros(n - n // 4), rng.exponential(2000.0, n // 4)])      # a spike plus a tail
    for v in (a, b, c):
        rng.shuffle(v)
    return np.stack([a, b, c])


def check(fit, ref):
    p, lb, n_iter, conv = ref
    assert fit.n_iter == n_iter and fit.converged == conv
    assert abs(fit.lower_bound - lb) < 1e-7 * max(1.0, abs(lb))
    np.testing.assert_allclose(fit.means, p['means'], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(fit.covariances, p['cov'], rtol=1e-8)
    np.testing.assert_allclose(fit.precisions_cholesky, p['pc'], rtol=1e-8)
    np.testing.assert_allclose(fit.degrees_of_freedom, p['dof'], rtol=1e-8)
    np.testing.assert_allclose(fit.mean_precision, p['mean_precision'], rtol=1e-8)
    np.testing.assert_allclose(fit.weight_concentration[0], p['a'], rtol=1e-8)
    np.testing.assert_allclose(fit.weight_concentration[1], p['b'], rtol=1e-8)
    np.testing.assert_allclose(fit.weights, gmm.weights(p), rtol=1e-8, atol=1e-300)


@pytest.mark.parametrize('n,K', [(300, 5), (1500, 3), (700, 8)])
def test_fit_matches_oracle_on_injected_responsibilities(n, K):
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    x = columns(n)
    rs = np.random.RandomState(5)
    resp = np.stack([np.stack([gmm.random_resp(rs, n, K) for _ in range(2)]) for _ in range(x.shape[0])])
    fits = DeviceBayesianGMM(n_components=K, n_init=2, device=DEV).fit_columns(x, resp_inj=resp)
    for c in range(x.shape[0]):
        check(fits[c], gmm.fit(x[c], list(resp[c])))


def test_fit_with_device_drawn_responsibilities():
    """The Philox initialisation: the same uniforms through dpc_fill, normalised on the host, feed the oracle."""
    from dp_cgans_amd import ops
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    n, K, n_init = 400, 5, 3
    x = columns(n)[:2]
    model = DeviceBayesianGMM(n_components=K, n_init=n_init, device=DEV, seed=21)
    fits = model.fit_columns(x)
    rng = ops.new_rng(21 ^ 0x6d6d, DEV)
    for c in range(2):
        resp = []
        for j in range(n_init):
            u = torch.empty(n, K, dtype=torch.float64, device=DEV)
            ops.fill(u, 4, rng, 0x4000 + c * n_init + j)
            u = u.cpu().numpy()
            resp.append(u / u.sum(axis=1)[:, None])
        check(fits[c], gmm.fit(x[c], resp))


def test_max_iter_stops_unconverged_fits():
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    x = columns(500)[:1]
    rs = np.random.RandomState(1)
    resp = gmm.random_resp(rs, 500, 5)[None, None]
    fit = DeviceBayesianGMM(n_components=5, n_init=1, max_iter=7, device=DEV).fit_columns(x, resp_inj=resp)[0]
    check(fit, gmm.fit(x[0], [resp[0, 0]], max_iter=7))
    assert fit.n_iter == 7 and not fit.converged


def test_rejects_bad_input():
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    m = DeviceBayesianGMM(device=DEV)
    with pytest.raises(ValueError):
        m.fit_columns(np.array([[1.0, np.nan, 2.0, 3.0, 4.0, 5.0]]))
    with pytest.raises(ValueError):
        m.fit_columns(np.array([[1.0, 2.0, 3.0]]))            # fewer rows than components
    assert m.fit_columns(np.zeros((0, 10))) == []


def test_transformer_fit_uses_the_device_mixtures():
    """DataTransformer.fit end to end: spans, valid components and the encode table come from the device fit."""
    import pandas as pd
    from dp_cgans_amd.data_transformer import DataTransformer
    n = 600
    x = columns(n)
    df = pd.DataFrame({'a': x[0], 'b': x[1], 'k': np.where(x[0] > 5, 'hi', 'lo')})
    t = DataTransformer(device=DEV, seed=3)
    t.fit(df, discrete_columns=['k'])
    infos = t._column_transform_info_list
    assert [i.column_type for i in infos] == ['continuous', 'continuous', 'discrete']
    for i in infos[:2]:
        assert i.output_dimensions == 1 + int(i.transform_aux.sum()) >= 2
        assert i.transform.converged and abs(i.transform.weights.sum() - 1) < 1e-12
    enc = t.transform(df)
    assert enc.shape == (n, t.output_dimensions) and np.abs(enc[:, 0]).max() <= 0.99
    back = t.inverse_transform(enc)
    assert list(back.columns) == ['a', 'b', 'k'] and (back['k'] == df['k']).all()


def test_two_valued_column():
    """A column with two distinct values (components collapse onto spikes; covariances fall back to
    reg_covar): same iterations and parameters as the oracle.  (A strictly constant column has a zero
    covariance prior; its dead components' covariances are nk-sized rounding residue in sklearn as
    well, so there is nothing to pin there.)"""
    from dp_cgans_amd.gmm_fit import DeviceBayesianGMM
    n, K = 256, 4
    rng = np.random.default_rng(4)
    x = (rng.integers(0, 2, n).astype(np.float64) * 7.0)[None]
    rs = np.random.RandomState(9)
    resp = gmm.random_resp(rs, n, K)[None, None]
    fit = DeviceBayesianGMM(n_components=K, n_init=1, device=DEV).fit_columns(x, resp_inj=resp)[0]
    check(fit, gmm.fit(x[0], [resp[0, 0]]))
    const = DeviceBayesianGMM(n_components=K, n_init=1, device=DEV).fit_columns(np.full((1, n), 3.25))[0]
    assert np.isfinite(const.means).all() and abs(const.weights.sum() - 1) < 1e-12     # runs, stays finite
