"""Timing of the generator's Linear layers at T1M: dense product over [dense | cond] columns against the
row-gather form (dense columns on the matrix cores + hot columns of W added in the epilogue)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
B, C = 8192, 276


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for F, Kd in [(256, 128), (256, 384), (335, 640)]:
    K = Kd + C
    x = torch.randn(B, 916, device=dev)[:, :K]
    W = torch.randn(F, K, device=dev)
    b = torch.randn(F, device=dev)
    out = torch.empty(B, (F + 3) // 4 * 4, device=dev)[:, :F]
    sel = torch.stack([torch.randint(0, C // 2, (B,)), torch.randint(C // 2, C, (B,))], 1).to(torch.int32).to(dev)
    WcT = torch.zeros(C, (F + 3) // 4 * 4, device=dev)[:, :F]
    ops.transpose_slices([(W[:, Kd:], WcT)])
    t_dense = timed(lambda: ops.gemm(x, W, out, trans_b=True, bias=b))
    t_part = timed(lambda: ops.gemm(x[:, :Kd], W[:, :Kd], out, trans_b=True, bias=b))
    t_gath = timed(lambda: ops.gemm_gather(x[:, :Kd], W[:, :Kd], out, WcT, sel, bias=b))
    print('F=%d Kd=%d: dense K=%d %.1f us | dense-part only %.1f us | gather %.1f us' % (F, Kd, K, t_dense, t_part, t_gath))
