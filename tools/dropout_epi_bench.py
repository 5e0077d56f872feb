"""The LeakyReLU + Dropout epilogue of the discriminator's hidden layers alone: the same product with Philox keep
bits, with a given keep mask (no draw in the epilogue) and with a ReLU epilogue, launches back to back in one graph.
usage: dropout_epi_bench.py [M [N [K]]]   (defaults: the T1M discriminator update's second layer, 3072 x 256 x 256)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
M, N, K = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 3072), (2, 256), (3, 256)))
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
C = torch.empty(M, N, device=dev)
aux = torch.empty(M, N, device=dev)
keep = (torch.rand(M, N, device=dev) < 0.5).to(torch.uint8)
rng = ops.new_rng(5, dev)
reps = 20


def timed(name, fn):
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
    print('%-34s %8.2f us' % (name, e0.elapsed_time(e1) * 1e3 / (5 * reps)), flush=True)


for sk in (1, 0):
    print('splitk =', sk)
    timed('relu epilogue', lambda: ops.gemm(A, W, C, trans_b=True, bias=b, epi=ops.EPI_RELU, splitk=sk))
    timed('leaky + dropout, keep mask given', lambda: ops.gemm(A, W, C, trans_b=True, bias=b, epi=ops.EPI_LEAKY_DROPOUT, aux_out=aux, keep=keep, splitk=sk))
    timed('leaky + dropout, Philox keep bits', lambda: ops.gemm(A, W, C, trans_b=True, bias=b, epi=ops.EPI_LEAKY_DROPOUT, aux_out=aux, rng=rng, site=3, splitk=sk))
