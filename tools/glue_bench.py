"""Stand-alone timing of the step's non-GEMM kernels at their in-step sizes (T1M, batch 8192): HIP events,
50 repetitions each, GB/s on the kernel's algorithmic bytes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd.data_sampler import DataSampler
from dp_cgans_amd.engine import StepEngine
from dp_cgans_amd.synthetic import build_encoded_table

dev = torch.device('cuda:0')
table, span = build_encoded_table('T1M', dev, n_rows=200000)
eng = StepEngine(span, 128, (256, 256), (256, 256), 8192, 8, dev, private=True, dp_mode='dpsgd', seed=0)
eng.sampler = DataSampler(table, span, device=dev, seed=1)
eng.train_step(1)
torch.cuda.synchronize()
B, Dd, n = eng.B, eng.Dd, eng.n


def timed(name, fn, nbytes, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print('%-28s %8.1f us  %7.0f GB/s' % (name, us, nbytes / us / 1e3))


c = eng.cg
x, y = c.fake, c.X[:, :Dd]
dy = torch.randn(B, Dd, device=dev)
dx = torch.zeros(B, Dd, device=dev)
timed('activate_fwd', lambda: ops.apply_activate_fwd(x, y, eng.spans, rng=eng.rng), 8.0 * B * Dd)
timed('activate_cond_loss_fwd', lambda: ops.activate_cond_loss_fwd(x, y, eng.spans, c.cond, c.mask, eng.col2d, eng.col2c, eng.nd,
                                                                   dx, eng.part_loss, rng=eng.rng), 4.0 * B * (3 * Dd + eng.nd + eng.C))
timed('activate_bwd', lambda: ops.apply_activate_bwd(y, dy, dx, eng.spans), 12.0 * B * Dd)
timed('cond_loss', lambda: ops.cond_loss(x, c.cond, c.mask, eng.col2d, eng.col2c, eng.nd, None, dx, eng.part_loss, accumulate=True,
                                         batch_total=B), 4.0 * B * (3 * Dd + eng.nd))
timed('hot_sort', lambda: ops.hot_sort(c.sel, eng.C, c.order, c.sid, c.n_items), 8.0 * 2 * B)
if len(sys.argv) > 1:
    sys.exit(0)
