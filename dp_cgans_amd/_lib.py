"""ctypes loader for libdpcgan_hip.so (the C-ABI declared in include/dpcgan_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C dp_cgans_amd/csrc``.  There
is NO fallback: if it is missing or fails to load, importing the kernels raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdpcgan_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dpcgan_hip.h')

_CTYPES = {
    'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32,
    'uint32_t': ctypes.c_uint32, 'uint64_t': ctypes.c_uint64, 'float': ctypes.c_float,
    'double': ctypes.c_double,
}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    with open(path) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', ' ', src, flags=re.M)
    src = src.replace('extern "C" {', ' ')
    protos = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(dpc_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if '*' in ret else _CTYPES[ret.replace('const', '').strip()]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace('const', '').split()[0]
                    argtypes.append(_CTYPES[ty])
        protos[name] = (restype, argtypes)
    return protos


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'dp_cgans_amd: %s not found. Build it first (python -c "import __graft_entry__ as g; '
                'g.build()" or make -C dp_cgans_amd/csrc). There is no CPU fallback.' % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes

    # When set to a list, every call is appended as (name, args) - bench.py uses it to issue the launches of one
    # training step again (replay) for the per-kernel roofline figures.  Measurement aid only.
    log = None

    def replay(self, entry, stream):
        """Issue a logged launch again on ``stream`` (every launching entry point takes the stream last)."""
        name, args = entry[0], entry[1]
        log, self.log = self.log, None
        try:
            self.call(name, *(args[:-1] + (stream,)))
        finally:
            self.log = log

    # Launch-order control for the engine's side branches (engine.StepEngine._Fork): while `defer` is a list, calls are
    # recorded there instead of issued; `after`, when set, is called with the stream argument of every issued call.
    defer = None
    after = None

    def call(self, name, *args):
        if self.defer is not None:
            self.defer.append((name, args))
            return
        if self.log is not None:
            self.log.append((name, args))
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.dpc_last_error()
            raise RuntimeError('%s failed (%d): %s' % (name, rc, msg.decode() if msg else ''))
        if self.after is not None and args:
            self.after(args[-1])


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
