"""Tile x split-K sweep for the large-K GEMMs of the T1M step (plain epilogue, beta = 0; the memset of a
split product is included, a deferred bias/activation pass is not)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd._lib import lib

dev = torch.device('cuda:0')
SHAPES = [
    ('G1 = W1 W1^T NT', 256, 256, 4888, False, True),          # round 3: the step's split products at their current shapes
    ('D fwd l0 x  NT', 2048, 256, 2816, False, True),
    ('G D fwd l0  NT', 1024, 256, 2816, False, True),
    ('P1 u^T u    TN', 256, 256, 1024, True, False),
    ('D fwd l0   NT', 3072, 256, 4888, False, True),
    ('D dW1      TN', 256, 4888, 3072, True, False),
    ('GP v1      NT', 1024, 256, 4888, False, True),
    ('G dWout    TN', 335, 916, 8192, True, False),
    ('G dW1      TN', 256, 660, 8192, True, False),
    ('G dW0      TN', 256, 404, 8192, True, False),
    ('G fwd out  NT', 8192, 335, 916, False, True),
    ('G dHg      NN', 8192, 512, 335, False, False),
    ('GP g0      NN', 1024, 4888, 256, False, False),
]
for tag, M, N, K, ta, tb in SHAPES:
    def mat(r, c):
        return torch.randn(r, (c + 3) // 4 * 4, device=dev)[:, :c]
    A = mat(K, M) if ta else mat(M, K)
    B = mat(N, K) if tb else mat(K, N)
    C = torch.zeros(M, N, device=dev)
    best = {}
    for tile in (64,):
        line = []
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if sp > 1 and K // sp < 128:
                continue
            for _ in range(2):
                ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=sp, tile=tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=sp, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            line.append('%d:%.0f' % (sp, us))
            if tile not in best or us < best[tile][0]:
                best[tile] = (us, sp)
        print('%-14s M=%5d N=%5d K=%5d tile %-5d  %s' % (tag, M, N, K, tile, '  '.join(line)), flush=True)
    for _ in range(2):
        ops.gemm(A, B, C, trans_a=ta, trans_b=tb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(A, B, C, trans_a=ta, trans_b=tb)
    e1.record()
    torch.cuda.synchronize()
    fl = 2.0 * M * N * K
    print('   auto %.0f us | best ' % (e0.elapsed_time(e1) * 100) + ', '.join('t%d sp%d %.0f us (%.0f TF/s)' % (t, b[1], b[0], fl / b[0] / 1e6) for t, b in best.items()), flush=True)
