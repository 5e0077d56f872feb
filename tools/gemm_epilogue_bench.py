"""What the fused epilogues of the generator's Linear layers cost (T1M shapes): the plain product against the same
product with bias, with the two-hot cond gather, and with the BatchNorm column statistics.  Graph-replayed, HIP events."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
C = 276


def timed(fn, reps=50):
    for _ in range(3):
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


for tag, M, N, K in [('G l0', 8192, 256, 128), ('G l1', 8192, 256, 384), ('G out', 8192, 335, 640)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    Cm = torch.zeros(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    gat = torch.randn(C, N, device=dev)
    gidx = torch.randint(0, C, (M, 2), device=dev, dtype=torch.int32)
    stat = torch.zeros(2, N, dtype=torch.float64, device=dev)
    cases = [('plain', dict()), ('bias', dict(bias=bias)), ('gather', dict(bias=bias, gat=gat, gidx=gidx)),
             ('colstat', dict(bias=bias, colstat=stat)), ('gather+colstat', dict(bias=bias, gat=gat, gidx=gidx, colstat=stat))]
    line = []
    for name, kw in cases:
        us = timed(lambda: ops.gemm(A, B, Cm, trans_b=True, **kw))
        line.append('%s %5.1f us' % (name, us))
    print('%-6s M=%5d N=%4d K=%4d  %s   (plain = %.1f TF/s)' % (tag, M, N, K, '  '.join(line), 0))
