"""Where a product's time goes, from INSIDE the kernel: the measurement build of gemm.hip (tools/gemm_stamp.sh) leaves
the shader-cycle counter at five points of every block's life and the 100 MHz wall clock at its two ends.
Per shape: shader cycles of prologue / K loop (per K-tile) / epilogue issue / store acknowledge, the clock the chip
ran at (cycles per wall nanosecond), and the kernel's span seen from inside (first block start - last block end)."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dp_cgans_amd import _lib
_lib.LIB_PATH = os.environ.get('DPC_STAMP_LIB') or os.path.join(ROOT, 'tools', '_stamp', 'libdpcgan_stamp.so')
import numpy as np
import torch
from dp_cgans_amd import ops

dev = torch.device('cuda:0')
L = _lib.lib()
L.cdll.dpc_debug_stamp_buffer.argtypes = [ctypes.c_void_p]


def run(tag, M, N, K, ta=False, tb=True, reps=30, **kw):
    def mat(r, c):
        return torch.randn(r, (c + 3) // 4 * 4, device=dev)[:, :c]
    A = mat(K, M) if ta else mat(M, K)
    B = mat(N, K) if tb else mat(K, N)
    C = torch.zeros(M, N, device=dev)
    nb = 4 * ((M + 63) // 64) * ((N + 63) // 64) * 64
    st = torch.zeros(nb * 16, dtype=torch.int64, device=dev)
    assert L.cdll.dpc_debug_stamp_buffer(None) == 0
    for _ in range(5):
        ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    assert L.cdll.dpc_debug_stamp_buffer(ctypes.c_void_p(st.data_ptr())) == 0
    for _ in range(3):          # the last run's stamps stay
        st.zero_()
        ops.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    torch.cuda.synchronize()
    assert L.cdll.dpc_debug_stamp_buffer(None) == 0
    s = st.cpu().numpy().reshape(-1, 16)
    s = s[s[:, 0] != 0]
    nblk = len(s)
    pro, loop, epi, ack = (s[:, 1] - s[:, 0]), (s[:, 2] - s[:, 1]), (s[:, 3] - s[:, 2]), (s[:, 4] - s[:, 3])
    tot = s[:, 4] - s[:, 0]
    wall = (s[:, 6] - s[:, 5]) * 10.0          # ns
    ktiles = max(K // 32, 1)
    span = (s[:, 6].max() - s[:, 5].min()) * 10.0
    ghz = np.median(tot[wall > 0] / wall[wall > 0])
    med = lambda x: float(np.median(x))
    print('%-18s M=%5d N=%5d K=%5d blocks %5d | launch %6.2f us, inside span %6.2f us | block: total %6.0f cyc '
          '(%5.2f us) = prologue %5.0f + loop %6.0f (%5.0f per K-tile) + epilogue %5.0f + store-ack %5.0f | clock %.2f GHz'
          % (tag, M, N, K, nblk, us, span / 1e3, med(tot), med(wall) / 1e3, med(pro), med(loop), med(loop) / ktiles,
             med(epi), med(ack), ghz))
    print('%-18s   epilogue split: loop end -> tile_out %5.0f, first tile_out %5.0f (entry -> loads + values %5.0f, -> '
          'activation %5.0f, -> 8 stores %5.0f, second half %5.0f), rest %5.0f cycles'
          % ('', med(s[:, 7] - s[:, 2]), med(s[:, 8] - s[:, 7]), med(s[:, 9] - s[:, 7]), med(s[:, 10] - s[:, 9]),
             med(s[:, 11] - s[:, 10]), med(s[:, 8] - s[:, 11]), med(s[:, 3] - s[:, 8])))
    # start-time spread of the blocks (dispatch ramp) and the last block's finish
    start = (s[:, 5] - s[:, 5].min()) * 10.0
    print('%-18s   block starts: median +%.2f us, 90 %% +%.2f us, last +%.2f us' % ('', np.median(start) / 1e3,
                                                                                  np.percentile(start, 90) / 1e3, start.max() / 1e3))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'loop':         # the K loop only (instruction-class experiments: DPC_EXP builds)
        print(os.path.basename(_lib.LIB_PATH))
        run('lone block', 64, 64, 2048, tile=64, splitk=1)
        run('2 blocks/CU', 8192, 256, 2048, tile=64, splitk=1)
        sys.exit(0)
    run('lone block', 64, 64, 256, tile=64, splitk=1)
    run('lone block', 64, 64, 2048, tile=64, splitk=1)
    run('1 block/CU', 4096, 256, 512, tile=64, splitk=1)
    run('2 blocks/CU', 8192, 256, 128, tile=64, splitk=1)
    run('2 blocks/CU', 8192, 256, 384, tile=64, splitk=1)
    run('2 blocks/CU', 8192, 256, 640, tile=64, splitk=1)
    run('2 blocks/CU', 8192, 256, 2048, tile=64, splitk=1)
    run('4 blocks/CU', 16384, 256, 512, tile=64, splitk=1)
    run('G dHg NN', 8192, 512, 335, tb=False, tile=64, splitk=1)
    run('G dW1 TN split', 256, 660, 8192, ta=True, tb=False)
    run('128 tile', 4096, 4096, 4096, tile=128, splitk=1, reps=5)
