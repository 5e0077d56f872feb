"""A LONE 64x64 block walking a long K (one wave per SIMD on one CU, nothing else on the chip): the run the SQ
counters of the K loop are collected on:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python3 tools/gemm_lone.py 64 64 16384
(a counter pass of its own: no trace domain beside --pmc)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 64, 16384)))
A = torch.randn(M, K, device=dev)
B = torch.randn(N, K, device=dev)
C = torch.zeros(M, N, device=dev)
for _ in range(20):
    ops.gemm(A, B, C, trans_b=True, tile=64, splitk=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(A, B, C, trans_b=True, tile=64, splitk=1)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / 20
print('M=%d N=%d K=%d: %.1f us per launch, %.3f us per K-tile' % (M, N, K, us, us / (K / 32)))
