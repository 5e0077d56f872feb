"""RDP accountant (dp_cgans_amd/rdp_accountant.py) against the grid captured from the reference's
functions/rdp_accountant.py:341-360,594-625 (tools/capture_goldens.py: capture_rdp).  CPU only."""
import numpy as np
import pytest

from golden_util import load
from dp_cgans_amd.rdp_accountant import compute_rdp, get_privacy_spent


def test_grid_matches_reference():
    z = load('rdp.npz')
    orders = z['orders'].tolist()
    for q, sigma, steps, eps, opt, r0, r50, r98 in z['grid'].tolist():
        rdp = compute_rdp(q=q, noise_multiplier=sigma, steps=int(steps), orders=orders)
        np.testing.assert_allclose([rdp[0], rdp[50], rdp[98]], [r0, r50, r98], rtol=1e-12, atol=0)
        e, d, o = get_privacy_spent(orders, rdp, target_delta=2e-6)
        np.testing.assert_allclose(e, eps, rtol=1e-12, atol=0)
        assert d == 2e-6 and o == opt


def test_argument_errors_match_reference():
    orders = [2.0, 3.0]
    rdp = compute_rdp(0.1, 1.0, 5, orders)
    with pytest.raises(ValueError):
        get_privacy_spent(orders, rdp)
    with pytest.raises(ValueError):
        get_privacy_spent(orders, rdp, target_eps=1.0, target_delta=1e-5)
    eps, delta, _ = get_privacy_spent(orders, rdp, target_eps=1.0)
    assert eps == 1.0 and 0 < delta <= 1
    assert compute_rdp(0.0, 1.0, 5, 2.0) == 0
    assert compute_rdp(1.0, 2.0, 3, 4.0) == 3 * 4.0 / 8.0
