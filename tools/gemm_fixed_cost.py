"""Where a product's fixed cost goes: chains of launches replayed from one HIP graph - a trivial kernel (fill of 64
floats), a one-block product, full grids with one K-tile, and the same with beta = 1 / bias."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')


def timed(fn, reps=100):
    for _ in range(2):
        fn()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        gr.capture_begin()
        for _ in range(reps):
            fn()
        gr.capture_end()
    torch.cuda.current_stream().wait_stream(st)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


rng = ops.new_rng(1, dev)
tiny = torch.empty(64, device=dev)
print('trivial kernel (fill 64 floats)      %6.2f us' % timed(lambda: ops.fill(tiny, ops.FILL_UNIFORM, rng, 1)))
for M, N, K in [(64, 64, 32), (64, 64, 256), (8192, 256, 32), (8192, 256, 64), (8192, 256, 128), (8192, 256, 256), (8192, 256, 512),
                (16384, 256, 128), (32768, 256, 128)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    C = torch.zeros(M, N, device=dev)
    us = timed(lambda: ops.gemm(A, B, C, trans_b=True, tile=64, splitk=1))
    print('M=%5d N=%4d K=%4d tiles %4d       %6.2f us  %5.1f TF/s' % (M, N, K, (M // 64) * (N // 64), us, 2e-6 * M * N * K / us))
