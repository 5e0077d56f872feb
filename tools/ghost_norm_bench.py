"""dpc_persample_clip_all_gram at the T1M step's shapes (n = 1024 examples, widths 256, K0x = 2816), graph-replayed:
with / without the cond-id counts and the implicit penalty row."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
n, d, K0x, pac, C = 1024, 256, 2816, 8, 276
if len(sys.argv) > 1:
    n = int(sys.argv[1])


def timed(fn, reps=20):
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


X = torch.randn(2 * n, K0x, device=dev)
X3 = torch.randn(3 * n, K0x, device=dev)
Dl = [torch.randn(3 * n, d, device=dev) * 0.1 for _ in range(2)]
H = [torch.randn(3 * n, d, device=dev) for _ in range(2)]
rs = torch.randn(3 * n, 1, device=dev)
zpre = torch.randn(2 * n, d, device=dev)
f = torch.randn(n, device=dev)
q = torch.rand(n, device=dev, dtype=torch.float64)
fc = torch.empty(n, device=dev)
sel_a = torch.randint(0, C, (n * pac, 2), device=dev, dtype=torch.int32)
sel_b = torch.randint(0, C, (n * pac, 2), device=dev, dtype=torch.int32)
norm2 = torch.empty(n, dtype=torch.float64, device=dev)
rowc, rowcw, nrm = torch.empty(3 * n, device=dev), torch.empty(3 * n, device=dev), torch.empty(n, device=dev)
roww = torch.randn(3 * n, device=dev)
layers_g = [(Dl[0], X), (Dl[1], H[0]), (rs, H[1])]
layers_d = [(Dl[0], X3), (Dl[1], H[0]), (rs, H[1])]
by = 4.0 * (3 * n * (2 * d + 1) + 2 * n * K0x + 3 * n * 2 * d + 4 * n * d)
for name, fn in [
    ('gram + ids', lambda: ops.persample_clip_all(layers_g, n, 1.0, 3.0, norm2, rowc, roww=roww, rowc_w=rowcw, norm_out=nrm, zpre=zpre,
                                                  gp_f=f, gp_q=q, fc_out=fc, clear_zpre=True, sel_a=sel_a, sel_b=sel_b, pac=pac)),
    ('gram', lambda: ops.persample_clip_all(layers_g, n, 1.0, 3.0, norm2, rowc, roww=roww, rowc_w=rowcw, norm_out=nrm, zpre=zpre,
                                            gp_f=f, gp_q=q, fc_out=fc, clear_zpre=True)),
    ('direct (3 K0-wide rows)', lambda: ops.persample_clip_all(layers_d, n, 1.0, 3.0, norm2, rowc, roww=roww, rowc_w=rowcw, norm_out=nrm)),
]:
    us = timed(fn)
    print('%-26s %6.1f us   %.0f GB/s of %.1f MB (gram form bytes)' % (name, us, by / us / 1e3, by / 1e6))
