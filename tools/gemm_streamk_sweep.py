"""Stream-K block-count sweep for the GEMM shapes of the T1M step, with a correctness check against
torch (fp64 reference).  -1 = automatic split-K (XCD-aware block order), 0 = automatic stream-K."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from dp_cgans_amd import ops
from dp_cgans_amd._lib import lib
from gemm_bench import SHAPES

dev = torch.device('cuda:0')
rng = ops.new_rng(1, dev)
tot = {}
for tag, M, N, K, ta, tb, epi in SHAPES:
    def mat(r, c):
        return torch.randn(r, (c + 3) // 4 * 4, device=dev)[:, :c]
    A = mat(K, M) if ta else mat(M, K)
    B = mat(N, K) if tb else mat(K, N)
    C = torch.zeros(M, N, device=dev)
    aux = torch.rand(M, N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)
    kw = {}
    if epi == 1:      # keep mask injected so that the result can be checked
        keep = (torch.rand(M, N, device=dev) < 0.5).to(torch.uint8)
        kw = dict(epi=1, aux_out=torch.empty(M, N, device=dev), keep=keep, bias=bias)
    elif epi == 2:
        kw = dict(epi=2, aux_in=aux)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    if epi == 1:
        pre = ref + bias.double()
        ref = pre * torch.where(pre > 0, 1.0, 0.2) * keep.double() * 2.0
    elif epi == 2:
        ref = ref * aux.double()
    line = []
    for nb in (-1, 0, 512, 768):
        lib().cdll.dpc_gemm_streamk_config(nb)
        for _ in range(2):
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
        err = float((C.double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, (tag, nb, err)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot[nb] = tot.get(nb, 0.0) + us
        line.append('%5d:%6.1f' % (nb, us))
    print('%-14s M=%5d N=%5d K=%5d  %s' % (tag, M, N, K, ' '.join(line)), flush=True)
lib().cdll.dpc_gemm_streamk_config(0)
print('sums: ' + ' '.join('%d:%.0f' % kv for kv in tot.items()))
