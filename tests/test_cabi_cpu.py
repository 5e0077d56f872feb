"""CPU-only: the C-ABI library loads and exports every symbol include/dpcgan_hip.h declares
(no compute calls here: there is no GPU in this container)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dp_cgans_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build the library first: python __graft_entry__.py'
    protos = _lib.parse_header()
    assert len(protos) >= 25
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name
    L = _lib.lib()
    assert L.cdll.dpc_version() >= 100


def test_argument_validation_needs_no_gpu():
    """Rejected arguments return DPC_EINVAL before anything touches the device."""
    from dp_cgans_amd import _lib
    L = _lib.lib()
    rc = L.cdll.dpc_gemm_f32(4, 4, 4, 1.0, None, 4, 1, None, 4, 1, 0.0, None, 4, None, 0, None, 0, None, 0, None, 0,
                             None, 0, 0.2, 0, None, None)
    assert rc == -1
    assert b'null operand' in L.cdll.dpc_last_error()
    with pytest.raises(RuntimeError):
        L.call('dpc_adam_step', None, None, None, None, 10, 1e-3, 0.5, 0.9, 1e-8, 0.0, 0.0, 1.0, None, 1, None, None,
               0, 0.0, None, None, 0, None)


def test_product_package_never_imports_the_oracle():
    import re
    pkg = os.path.join(ROOT, 'dp_cgans_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
