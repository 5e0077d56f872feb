"""The small products of the discriminator's chains (K = 256 links, batch 1024 / 3072 pac groups): the 64x64 tile
against the 32x32 tile whose waves split K, HIP events, 50 repetitions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
for tag, M, N, K, ta, tb in [('D l1 fwd NT', 3072, 256, 256, False, True), ('D chain NN', 3072, 256, 256, False, False),
                             ('Tq NN', 1024, 256, 256, False, False), ('v chain NT', 1024, 256, 256, False, True),
                             ('P1 TN', 256, 256, 1024, True, False), ('dW2 TN', 256, 256, 3072, True, False),
                             ('G l0 NT', 8192, 256, 128, False, True)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    line = []
    for tile in (64, 32, 324):
        if tile != 64 and K > 1024:
            continue
        for _ in range(3):
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb, tile=tile)
        # 50 launches replayed as one HIP graph: eager launches of kernels this short are bound by the host
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            gr = torch.cuda.CUDAGraph()
            gr.capture_begin()
            for _ in range(50):
                ops.gemm(A, B, C, trans_a=ta, trans_b=tb, tile=tile)
            gr.capture_end()
        torch.cuda.current_stream().wait_stream(st)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        line.append('tile %d: %6.1f us %5.1f TF/s' % (tile, us, 2e-6 * M * N * K / us))
    print('%-12s M=%5d N=%4d K=%5d   %s' % (tag, M, N, K, '   '.join(line)))
