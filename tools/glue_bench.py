"""Stand-alone timing of the step's glue kernels at the T1M launch sizes (HIP events, 50 reps)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd.engine import StepEngine
from dp_cgans_amd.synthetic import schema_span_info

dev = torch.device('cuda:0')
span = schema_span_info('T1M')
eng = StepEngine(span, 128, (256, 256), (256, 256), 8192, 8, dev, private=True, dp_mode='dpsgd')
B, Dd, n, K0 = eng.B, eng.Dd, eng.n, eng.K0
x = torch.randn(B, Dd, device=dev)
y = torch.empty(B, Dd, device=dev)
dy = torch.randn(B, Dd, device=dev)
dx = torch.empty(B, Dd, device=dev)
cond = (torch.rand(B, eng.C, device=dev) < 0.01).float()
mask = (torch.rand(B, eng.nd, device=dev) < 0.02).int()


def timed(name, fn, bytes_):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    print('%-18s %7.1f us  %6.0f GB/s (algorithmic)' % (name, us, bytes_ / us / 1e3))


timed('activate_fwd', lambda: ops.apply_activate_fwd(x, y, eng.spans, rng=eng.rng), 8.0 * B * Dd)
timed('activate_bwd', lambda: ops.apply_activate_bwd(y, dy, dx, eng.spans), 12.0 * B * Dd)
timed('cond_loss', lambda: ops.cond_loss(x, cond, mask, eng.col2d, eng.col2c, eng.nd, None, dx, eng.part_loss, accumulate=True,
                                          batch_total=B), 4.0 * B * (2 * Dd + eng.nd))
timed('interpolate', lambda: ops.interpolate(eng.Xd[B:2 * B], eng.Xd[0:B], eng.Xd[2 * B:3 * B], n, K0, rng=eng.rng), 12.0 * n * K0)
timed('gp_norm', lambda: ops.gp_norm(eng.Xd3[2 * n:3 * n], 0.01, eng.pen_rows), 8.0 * n * K0)
z = torch.randn(B, 256, device=dev)
timed('colreduce<0>', lambda: ops.colreduce_partial(0, z, eng.part), 4.0 * B * 256)
timed('bn_relu_fwd', lambda: ops.bn_relu_fwd(z, eng.cd.Hg[:, 256:512], eng.cd.bn_mean[0], eng.cd.bn_invstd[0], eng.cd.bn_mean[0],
                                             eng.cd.bn_mean[0]), 8.0 * B * 256)
timed('fill z', lambda: ops.fill(eng.cd.Hg[:, 512:640], ops.FILL_NORMAL, eng.rng, 1, 0.0, 1.0), 4.0 * B * 128)
