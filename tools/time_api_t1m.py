#!/usr/bin/env python3
"""The public API at the T1M scale: DP_CGAN(...).fit(DataFrame of 1M rows x 50 columns) + sample().
Raw table as SURVEY 8(d): 20 continuous columns (mixtures of 1-3 Gaussians), 30 categorical string
columns (cardinality [2,3,4,6,8], Zipf).  Prints the wall time of fit (transformer fit + encode + sampler
tables + epochs) and of the step loop alone."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def raw_table(N, seed=1234):
    import pandas as pd
    rng = np.random.default_rng(seed)
    cols = {}
    for j in range(20):
        m = 1 + j % 3
        comp = rng.integers(m, size=N)
        cols['x%d' % j] = rng.normal(10.0 * comp, 1.0 + 0.25 * comp)
    cards = [2, 3, 4, 6, 8]
    for j in range(30):
        k = cards[j % 5]
        p = 1.0 / np.arange(1, k + 1)
        p /= p.sum()
        codes = rng.choice(k, size=N, p=p)
        cols['c%d' % j] = np.array(['c%d_%d' % (j, v) for v in range(k)], dtype=object)[codes]
    return pd.DataFrame(cols)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1000000)
    ap.add_argument('--epochs', type=int, default=1)
    a = ap.parse_args()
    from dp_cgans_amd import DP_CGAN
    t = time.perf_counter()
    df = raw_table(a.rows)
    print('raw table %d x %d built in %.1f s' % (df.shape[0], df.shape[1], time.perf_counter() - t), flush=True)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            m = DP_CGAN(epochs=a.epochs, batch_size=8192, pac=8, private=True, dp_mode='dpsgd', verbose=False)
            t = time.perf_counter()
            m.fit(df)
            fit_s = time.perf_counter() - t
            s = m._model
            print('fit: %.1f s total; step loop %.2f s for %d steps (%.3f ms per step, %.2f M rows/s)' % (
                fit_s, s.step_loop_seconds, s.steps_done, 1e3 * s.step_loop_seconds / max(s.steps_done, 1),
                8192 * s.steps_done / s.step_loop_seconds / 1e6), flush=True)
            t = time.perf_counter()
            out = m.sample(100000)
            print('sample(100000): %.2f s -> %s' % (time.perf_counter() - t, out.shape), flush=True)
        finally:
            os.chdir(cwd)


if __name__ == '__main__':
    main()
