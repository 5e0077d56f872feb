import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dp_cgans_amd import ops
dev = torch.device('cuda:0')
def run(M, N, K, ta, tb, reps=50):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    for _ in range(5): ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(A, B, C, trans_a=ta, trans_b=tb, splitk=1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N) in [(3072, 256), (8192, 256), (1024, 256), (64, 64)]:
    for K in [32, 64, 128, 256, 512, 1024, 2048, 4096]:
        t = run(M, N, K, False, True)
        print('NT M=%5d N=%4d K=%5d  %7.1f us  %6.1f TF/s  blocks=%d' % (M, N, K, t, 2.0 * M * N * K / t / 1e6, ((M + 63) // 64) * ((N + 63) // 64)))
# empty-ish kernel launch cost
x = torch.zeros(64, device=dev)
rng = ops.new_rng(1, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.state_advance(rng)
e1.record(); torch.cuda.synchronize()
print('tiny kernel back-to-back: %.2f us' % (e0.elapsed_time(e1) / 200 * 1e3))
