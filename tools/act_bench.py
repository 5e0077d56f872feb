"""The activation kernels alone at their in-step size (T1M schema, 8192 x 335): HIP events over back-to-back launches
inside one captured graph (no host gaps).  usage: act_bench.py [reps [T1M|T10M|TWIDE]]   (under rocprofv3 --pmc: the same launches)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd.data_sampler import DataSampler
from dp_cgans_amd.engine import StepEngine
from dp_cgans_amd.synthetic import build_encoded_table

dev = torch.device('cuda:0')
name = sys.argv[2] if len(sys.argv) > 2 else 'T1M'
table, span = build_encoded_table(name, dev, n_rows=200000 if name == 'T1M' else 20000)
eng = StepEngine(span, 128, (256, 256), (256, 256), 8192, 8, dev, private=True, dp_mode='dpsgd', seed=0)
eng.sampler = DataSampler(table, span, device=dev, seed=1)
eng.train_step(1)
torch.cuda.synchronize()
B, Dd = eng.B, eng.Dd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = eng.cg
x, y = c.fake, torch.empty(B, Dd, device=dev)
dx = torch.zeros(B, Dd, device=dev)
dy = torch.randn(B, Dd, device=dev)


def timed(name, fn, nbytes):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    print('%-28s %8.2f us  %7.0f GB/s' % (name, us, nbytes / us / 1e3), flush=True)


timed('copy (torch)', lambda: y.copy_(x), 8.0 * B * Dd)
timed('activate_fwd', lambda: ops.apply_activate_fwd(x, y, eng.spans, rng=eng.rng), 8.0 * B * Dd)
timed('activate_cond_loss_fwd', lambda: ops.activate_cond_loss_fwd(x, y, eng.spans, c.cond, c.mask, eng.col2d, eng.col2c, eng.nd,
                                                                   dx, eng.part_loss, rng=eng.rng), 4.0 * B * (3 * Dd + eng.nd + eng.C))
timed('activate_bwd', lambda: ops.apply_activate_bwd(y, dy, dx, eng.spans), 12.0 * B * Dd)
