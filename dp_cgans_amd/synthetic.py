"""Synthetic tables of the shapes BASELINE.json / SURVEY.md 8(d) name, and their encode on the device.

T1M   (config 3, seed 1234): 1 000 000 rows, 20 continuous + 30 categorical columns
T10M  (config 4, seed 1235): same schema, 10 000 000 rows
TWIDE (config 5, seed 1236): 200 000 rows, 400 continuous + 100 categorical columns
Continuous column j ~ equal-weight mixture of m_j = 1 + (j mod 3) Gaussians, means 10c, std 1 + .25c.
Categorical column j has cardinality k_j (T1M/T10M: [2,3,4,6,8][j mod 5]; TWIDE: [10,50,100,200][j mod 4])
with Zipf(s=1) probabilities.  As through the reference's public wrapper, each category
of a categorical column becomes its own binary column, and every binary column is a 2-way discrete
column of the synthesizer.

Instead of running sklearn's BayesianGaussianMixture fit (minutes to hours at these sizes, and listed
as a later row), the "fitted transformer" is written down directly from the generating
mixture: mu_k, sigma_k, weight 1/m.  The encode itself runs on the HIP kernel.
"""
import math

import numpy as np
import torch

from . import ops

SCHEMAS = {
    'T1M': dict(seed=1234, N=1000000, n_cont=20, n_cat=30, cards=[2, 3, 4, 6, 8]),
    'T10M': dict(seed=1235, N=10000000, n_cont=20, n_cat=30, cards=[2, 3, 4, 6, 8]),
    'TWIDE': dict(seed=1236, N=200000, n_cont=400, n_cat=100, cards=[10, 50, 100, 200]),
    'TINY': dict(seed=1237, N=5000, n_cont=4, n_cat=5, cards=[2, 3, 4, 6, 8]),
}


def schema_span_info(name):
    s = SCHEMAS[name]
    rows = []
    for j in range(s['n_cont']):
        rows += [(1, 0, 2), (1 + j % 3, 1, 2)]
    for j in range(s['n_cat']):
        rows += [(2, 1, 1)] * s['cards'][j % len(s['cards'])]
    return np.array(rows, dtype=np.int32)


def build_encoded_table(name, device, n_rows=None, chunk=1 << 20):
    """-> (encoded fp32 [N, Dd] on the device, span_info).  Raw columns are drawn with numpy
    default_rng(seed) chunk by chunk and encoded by dpc_encode_continuous / dpc_encode_onehot."""
    s = SCHEMAS[name]
    N = n_rows or s['N']
    span = schema_span_info(name)
    Dd = int(span[:, 0].sum())
    dev = torch.device(device)
    out = torch.empty(N, Dd, device=dev)
    rng_np = np.random.default_rng(s['seed'])
    rng = ops.new_rng(s['seed'], dev)
    tabs, valids = [], []
    for j in range(s['n_cont']):
        m = 1 + j % 3
        mu = 10.0 * np.arange(m)
        sg = 1.0 + 0.25 * np.arange(m)
        A = np.log(1.0 / m) + np.log(1.0 / sg) - 0.5 * math.log(2 * math.pi)
        tabs.append(torch.from_numpy(np.stack([mu, sg, 1.0 / sg, A])).to(dev))
        valids.append(torch.arange(m, dtype=torch.int32, device=dev))
    for r0 in range(0, N, chunk):
        r1 = min(N, r0 + chunk)
        nr = r1 - r0
        view = out[r0:r1]
        col = 0
        for j in range(s['n_cont']):
            m = 1 + j % 3
            comp = rng_np.integers(m, size=nr)
            x = rng_np.normal(10.0 * comp, 1.0 + 0.25 * comp)
            ops.encode_continuous(torch.from_numpy(x).to(dev), tabs[j], valids[j], out32=view, col_off=col, rng=rng,
                                  site=ops.SITE_ENCODE + j)
            col += 1 + m
        for j in range(s['n_cat']):
            k = s['cards'][j % len(s['cards'])]
            p = 1.0 / np.arange(1, k + 1)
            p /= p.sum()
            codes = rng_np.choice(k, size=nr, p=p)
            cd = torch.from_numpy(codes.astype(np.int32)).to(dev)
            for v in range(k):
                # binary column "category == v": one-hot over (is v, is not v)
                ops.encode_onehot((cd != v).to(torch.int32), 2, out32=view, col_off=col)
                col += 2
        ops.state_advance(rng)
        assert col == Dd
    return out, span
