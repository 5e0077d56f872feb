#!/usr/bin/env python3
"""Times the reference's own DP_CGAN.fit on the CPU of THIS container (CPU-baseline
plan item 1).  Imports /root/reference/src, so it only runs where the reference is mounted; it is
never used on the GPU box.  rows/s = batch x G-steps / wall time of epochs 1.. (epoch 0 excluded)."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_goldens import import_reference, REF   # gensim stub + sys.path


def run(cfg, threads, epochs):
    import numpy as np
    import pandas as pd
    import torch
    from dp_cgans import DP_CGAN
    from dp_cgans.synthesizers import dp_cgan as ref
    torch.set_num_threads(threads)
    np.random.seed(0)
    torch.manual_seed(0)
    df = pd.read_csv(os.path.join(REF, 'resources', 'example_tabular_data_UCIAdult.csv'))
    stamps = []
    _step = torch.optim.Adam.step
    state = {'n': 0}

    def step(self, closure=None):
        out = _step(self, closure)
        state['n'] += 1
        stamps.append(time.perf_counter())
        return out
    torch.optim.Adam.step = step
    try:
        model = DP_CGAN(epochs=epochs, cuda=False, verbose=False, **cfg)
        model.fit(df)
    finally:
        torch.optim.Adam.step = _step
    per_epoch = len(stamps) // epochs          # optimiser steps per epoch (D steps + 1 G step per batch)
    t = stamps[-1] - stamps[per_epoch - 1]     # epochs 1 .. end
    gsteps = (epochs - 1) * max(len(df) // cfg['batch_size'], 1)
    return cfg['batch_size'] * gsteps / t


def main():
    import_reference()
    cfgs = {
        'config1 (Adult, B=100, dims (128,128,128), d_steps 1, private=False)':
            dict(batch_size=100, generator_dim=(128, 128, 128), discriminator_dim=(128, 128, 128), private=False),
        # BASELINE config 2 proper (B=500 > N=200) makes the reference raise 'math domain error' in its RDP
        # accounting (sampling probability 2.5, dp_cgan.py:810-815), so private=True is timed at B=100
        'config2* (Adult, B=100, pac 10, dims (256,256), private=True; B=500 raises in the reference)':
            dict(batch_size=100, pac=10, private=True),
    }
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for name, cfg in cfgs.items():
                for threads in (1, 8):
                    r = run(cfg, threads, epochs=8)
                    print('%s threads=%d: %.0f rows/s' % (name, threads, r), flush=True)
        finally:
            os.chdir(cwd)


if __name__ == '__main__':
    main()
