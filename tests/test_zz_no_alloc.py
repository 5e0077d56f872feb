"""dpc_colsum_noalloc (the entry point that allocates nothing: its row-chunk partial sums live in a
file-scope device array of the library) against the existing pair dpc_colreduce_partial(mode 1) +
dpc_colreduce_finish(kind 1), bit for bit, through the C-ABI: two calls in a row, the second on a
smaller input (the scratch keeps the first call's contents), then the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pair(x, w, accumulate_into=None):
    from dp_cgans_amd import ops
    rows, F = x.shape
    part = torch.zeros(64 * F, dtype=torch.float64, device=DEV)
    out = torch.zeros(F, device=DEV) if accumulate_into is None else accumulate_into.clone()
    R = ops.colreduce_partial(1, x, part, w=w)
    ops.colreduce_finish(part, R, 1, F, 1, out, accumulate=accumulate_into is not None)
    return out


def test_colsum_noalloc_matches_the_workspace_pair_on_two_calls_in_a_row():
    from dp_cgans_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    big = torch.randn(8192, 1000, device=DEV, generator=g)[:, :997]          # rows padded, odd width
    wb = torch.rand(8192, device=DEV, generator=g)
    small = torch.randn(300, 70, device=DEV, generator=g)
    before = torch.cuda.memory_allocated(DEV)
    out_big = torch.full((997,), 7.0, device=DEV)
    out_small = torch.full((70,), -3.0, device=DEV)
    base = out_small.clone()
    mid = torch.cuda.memory_allocated(DEV)
    ops.colsum_noalloc(big, out_big, w=wb)
    ops.colsum_noalloc(small, out_small, accumulate=True)                    # smaller, right behind the first
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(DEV) == mid                           # nothing allocated through torch either
    assert torch.equal(out_big, _pair(big, wb))
    assert torch.equal(out_small, _pair(small, None, accumulate_into=base))
    # against the plain fp64 sum, to the fp32 bar
    ref = (big.double() * wb.double()[:, None]).sum(0)
    assert float((out_big.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert before <= mid


def test_colsum_noalloc_rejects_more_columns_than_its_scratch_holds():
    from dp_cgans_amd import ops
    x = torch.zeros(4, 8193, device=DEV)
    with pytest.raises(RuntimeError, match='8192'):
        ops.colsum_noalloc(x, torch.zeros(8193, device=DEV))
    ok = torch.ones(4, 8192, device=DEV)
    out = torch.zeros(8192, device=DEV)
    ops.colsum_noalloc(ok, out)
    assert torch.equal(out, torch.full((8192,), 4.0, device=DEV))


def test_colsum_noalloc_rejects_operands_that_are_not_on_the_current_device():
    """Through the bare C-ABI (the ops wrapper has its own device check): a host pointer, and - where a second GPU
    is visible - operands of another device than the current one."""
    import ctypes
    from dp_cgans_amd._lib import lib
    L = lib()
    x = torch.ones(8, 16, device=DEV)
    out = torch.zeros(16, device=DEV)
    host = torch.ones(8, 16)
    st = torch.cuda.current_stream(DEV).cuda_stream
    with pytest.raises(RuntimeError, match='colsum_noalloc'):
        L.call('dpc_colsum_noalloc', host.data_ptr(), 16, 8, 16, None, out.data_ptr(), 0, st)
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1):
            with pytest.raises(RuntimeError, match='current device'):
                L.call('dpc_colsum_noalloc', x.data_ptr(), 16, 8, 16, None, out.data_ptr(), 0,
                       torch.cuda.current_stream(1).cuda_stream)
    L.call('dpc_colsum_noalloc', x.data_ptr(), 16, 8, 16, None, out.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert torch.equal(out, torch.full((16,), 8.0, device=DEV))
