"""Steady-state rate of the GEMM inner loop by blocks per CU (long K, so fixed costs vanish)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
for tile in (64, 128):
    for layout in ('NT', 'NN', 'TN'):
        ta, tb = layout[0] == 'T', layout[1] == 'T'
        for M, N, K in ((1024, 1024, 16384), (2048, 1024, 16384), (2048, 2048, 16384), (4096, 2048, 16384), (4096, 4096, 8192)):
            A = torch.randn((K, M) if ta else (M, K), device=dev)
            B = torch.randn((N, K) if tb else (K, N), device=dev)
            C = torch.zeros(M, N, device=dev)
            for _ in range(2):
                ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=1, tile=tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=1, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 200
            blocks = ((M + tile - 1) // tile) * ((N + tile - 1) // tile)
            print('tile %3d %s M=%4d N=%4d K=%5d  %5d blocks (%.1f per CU)  %8.1f us  %6.1f TF/s  %.1f ns per k per block-slot' % (
                tile, layout, M, N, K, blocks, blocks / 256, us, 2.0 * M * N * K / us / 1e6, us * 1e3 / K / max(1, blocks / 256)), flush=True)
