#!/usr/bin/env python3
"""Training rows/s of DP_CGAN.fit on the Adult example for BASELINE configs 1 and 2 (one MI355X).
rows/s = batch x G-steps / wall time of the step loop (transformer fit / encode / sampler build excluded)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import pandas as pd
    from dp_cgans_amd import DP_CGAN
    df = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'adult.csv'))
    cfgs = [
        ('config1 Adult B=100 dims (128,128,128) private=False', dict(batch_size=100, generator_dim=(128, 128, 128), discriminator_dim=(128, 128, 128), private=False)),
        ('config2 Adult B=500 pac 10 private=True dp_mode=dpsgd', dict(batch_size=500, pac=10, private=True, dp_mode='dpsgd')),
        ('config2 Adult B=500 pac 10 private=True dp_mode=reference', dict(batch_size=500, pac=10, private=True, dp_mode='reference')),
        ('config2* Adult B=100 pac 10 private=True dp_mode=reference', dict(batch_size=100, pac=10, private=True, dp_mode='reference')),
    ]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for name, cfg in cfgs:
                m = DP_CGAN(epochs=300, **cfg)
                m.fit(df)
                s = m._model
                print('%s: %.0f rows/s (%.3f ms per D+G step, %d steps)' % (
                    name, cfg['batch_size'] * s.steps_done / s.step_loop_seconds, 1e3 * s.step_loop_seconds / s.steps_done, s.steps_done), flush=True)
        finally:
            os.chdir(cwd)


if __name__ == '__main__':
    main()
