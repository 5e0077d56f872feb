"""The noise-fused Adam launch on buffers beyond the Infinity Cache: the flat step, and the step that also moves the
cond slices of a TWIDE-shaped first layer (rows 256, pac 8 sub-rows of 19200 data + 18000 cond columns)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
rng = ops.new_rng(3, dev)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


rows, pac, Dd, C = 256, 8, 19200, 18000
row = Dd + C
n = rows * pac * row + 70000
p = torch.randn(n, device=dev) * 0.01
g, m, v = torch.randn(n, device=dev) * 1e-3, torch.zeros(n, device=dev), torch.zeros(n, device=dev)
t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
kw = dict(t_dev=t_dev, t_host=1, noise_scale=1e-3, rng=rng, site=ops.SITE_NOISE_D)
us = timed(lambda: ops.adam_step(p, g, m, v, 2e-4, 1e-6, **kw))
print('flat, noise from Philox:   %8.1f us  %6.0f GB/s (28 B per parameter, %d M parameters)' % (us, 28.0 * n / us / 1e3, n // 1000000))
us = timed(lambda: ops.adam_step(p, g, m, v, 2e-4, 1e-6, t_dev=t_dev, t_host=1))
print('flat, no noise:            %8.1f us  %6.0f GB/s' % (us, 28.0 * n / us / 1e3))
W1 = p[:rows * pac * row].view(rows, pac * row)
gT = torch.randn(pac * C, rows, device=dev) * 1e-3
pT = torch.zeros(pac * C, rows, device=dev)
sl = [(W1, s * row + Dd, C, gT[s * C:(s + 1) * C], pT[s * C:(s + 1) * C]) for s in range(pac)]
us = timed(lambda: ops.adam_step_slices(p, g, m, v, 2e-4, 1e-6, [(W1, row, Dd)], sl, **kw))
by = 28.0 * n + 8.0 * pac * C * rows
print('with the cond slices:      %8.1f us  %6.0f GB/s (%.2f GB)' % (us, by / us / 1e3, by / 1e9))

for what, gt_on, pt_on in (('slices, no addend, no copy', 0, 0), ('slices, addend only', 1, 0), ('slices, copy only', 0, 1)):
    sl2 = [(W, c0, nc, a if gt_on else None, b if pt_on else None) for W, c0, nc, a, b in sl]
    us = timed(lambda: ops.adam_step_slices(p, g, m, v, 2e-4, 1e-6, [(W1, row, Dd)], sl2, **kw))
    by = 28.0 * n + 4.0 * (gt_on + pt_on) * pac * C * rows
    print('%-26s %8.1f us  %6.0f GB/s' % (what + ':', us, by / us / 1e3))

# ---- the T1M discriminator's size (1.32 M parameters: cache-resident, latency-bound): a chain of 50 launches in one graph
del p, g, m, v, gT, pT, W1, sl
rows, pac, Dd, C = 256, 8, 335, 276
row = Dd + C
n = rows * pac * row + 67000
p = torch.randn(n, device=dev) * 0.01
g, m, v = torch.randn(n, device=dev) * 1e-3, torch.zeros(n, device=dev), torch.zeros(n, device=dev)
W1 = p[:rows * pac * row].view(rows, pac * row)
gT = torch.randn(pac * C, rows, device=dev) * 1e-3
pT = torch.zeros(pac * C, rows, device=dev)
sl = [(W1, s * row + Dd, C, gT[s * C:(s + 1) * C], pT[s * C:(s + 1) * C]) for s in range(pac)]


def chain(fn, reps=50):
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
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (5 * reps)


print('T1M size, flat:                 %6.2f us per launch' % chain(lambda: ops.adam_step(p, g, m, v, 2e-4, 1e-6, **kw)))
print('T1M size, with the cond slices: %6.2f us per launch' % chain(
    lambda: ops.adam_step_slices(p, g, m, v, 2e-4, 1e-6, [(W1, row, Dd)], sl, **kw)))
