"""Per-shape timing of the GEMMs one T1M training step issues (HIP events, 20 reps each)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
SHAPES = [  # (tag, M, N, K, trans_a, trans_b, epi)
    ('G fwd l0   NT', 8192, 256, 404, False, True, 0),
    ('G fwd l1   NT', 8192, 256, 660, False, True, 0),
    ('G fwd out  NT', 8192, 335, 916, False, True, 0),
    ('D fwd l0   NT', 3072, 256, 4888, False, True, 1),
    ('D fwd l1   NT', 3072, 256, 256, False, True, 1),
    ('D chain    NN', 3072, 256, 256, False, False, 2),
    ('GP g0      NN', 1024, 4888, 256, False, False, 0),
    ('D dW1      TN', 256, 4888, 3072, True, False, 0),
    ('GP v1      NT', 1024, 256, 4888, False, True, 2),
    ('D dW2      TN', 256, 256, 3072, True, False, 0),
    ('Dg fwd l0  NT', 1024, 256, 4888, False, True, 1),
    ('G dWout    TN', 335, 916, 8192, True, False, 0),
    ('G dHg      NN', 8192, 512, 335, False, False, 0),
    ('G dW1      TN', 256, 660, 8192, True, False, 0),
    ('G dX1      NN', 8192, 256, 256, False, False, 0),
    ('G dW0      TN', 256, 404, 8192, True, False, 0),
]
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    SHAPES = [('big NT', 4096, 4096, 4096, False, True, 0), ('big NN', 4096, 4096, 4096, False, False, 0),
              ('big TN', 4096, 4096, 4096, True, False, 0), ('mid NT', 8192, 1024, 1024, False, True, 0)]


def main():
    rng = ops.new_rng(1, dev)
    tot_ms, tot_fl = 0.0, 0.0
    from dp_cgans_amd._lib import lib
    TILES = [0]
    if len(sys.argv) > 1 and sys.argv[1] == 'tiles':
        TILES = [0, 64, 128]
    for tag, M, N, K, ta, tb, epi in [sh for sh in SHAPES for _ in TILES]:
        def mat(r, c):      # rows padded to 16 bytes, as the engine's buffers are
            return torch.randn(r, (c + 3) // 4 * 4, device=dev)[:, :c]
        A = mat(K, M) if ta else mat(M, K)
        B = mat(N, K) if tb else mat(K, N)
        C = torch.zeros(M, N, device=dev)
        aux = torch.ones(M, N, device=dev)
        kw = {}
        if epi == 1:
            kw = dict(epi=1, aux_out=aux, rng=rng, site=20, bias=torch.zeros(N, device=dev))
        elif epi == 2:
            kw = dict(epi=2, aux_in=aux)
        tile = TILES[0]
        TILES.append(TILES.pop(0))
        kw['tile'] = tile
        tag = '%s t%-5d' % (tag, tile)
        for _ in range(3):
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        print('%-22s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TF/s' % (tag, M, N, K, ms * 1e3, fl / ms / 1e9))
    print('sum %.1f us, %.1f TF/s aggregate' % (tot_ms * 1e3, tot_fl / tot_ms / 1e9))


if __name__ == '__main__':
    main()
