"""ctypes binding of libgeobi_hip.so, derived from include/geobi_hip.h.

The prototypes are parsed from the public header so the binding cannot drift from the
declared C ABI.  There is no CPU fallback: if the library is missing or a call fails the
caller gets an exception (``GeobiError``).
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), 'include', 'geobi_hip.h')
# GEOBI_LIB: load another build of the same ABI (A/B timing of two builds on one box)
LIB_PATH = os.environ.get('GEOBI_LIB') or os.path.join(_PKG, 'libgeobi_hip.so')


class GeobiError(RuntimeError):
    pass


_CTYPES = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32, 'size_t': ctypes.c_size_t,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'\b(int|size_t|const char\s*\*)\s+(geobi_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t}.get(ret, ctypes.c_char_p)
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    base = a.replace('const ', '').split('*')[0].strip()
                    if base in ('double', 'int64_t') and not a.startswith('const'):
                        argtypes.append(ctypes.POINTER(_CTYPES[base]))   # host out-parameters
                    else:
                        argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split('*')[-1].strip())
                else:
                    t, n = a.rsplit(' ', 1)
                    argtypes.append(_CTYPES[t.replace('const ', '').strip()])
                    argnames.append(n)
        protos[name] = (restype, argtypes, argnames)
    return protos


_lib = None


def lib():
    """The loaded library; raises GeobiError (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GeobiError('libgeobi_hip.so is not built (%s); run `python -c "import __graft_entry__ as g; '
                             'g.build()"` or geobi_gnn_amd/csrc/build.sh' % LIB_PATH)
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise GeobiError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (restype, argtypes, _) in parse_header().items():
            fn = getattr(handle, name, None)
            if fn is None:
                # an OLDER build named through GEOBI_LIB for a same-box A/B may lack entry points added since (calling one
                # then fails in ctypes); the product library must export everything the header declares
                if os.environ.get('GEOBI_LIB') and os.environ.get('GEOBI_LIB_OLDER') == '1':
                    continue
                raise GeobiError('libgeobi_hip.so does not export %s (declared in include/geobi_hip.h)' % name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().geobi_last_error()
        raise GeobiError('%s failed: %s' % (what or 'geobi call', msg.decode() if msg else 'unknown error'))


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise GeobiError('non-contiguous tensor handed to the C ABI')
    return t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device.

    torch.cuda.current_stream() builds a Python Stream object (~9 us); the raw getter is ~0.3 us and
    this sits on every C-ABI call."""
    return _raw_stream(torch.cuda.current_device())


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:          # pragma: no cover - older torch
    def _raw_stream(device_index):
        return torch.cuda.current_stream(device_index).cuda_stream


def require_device(t, what='tensor'):
    if not t.is_cuda:
        raise GeobiError('%s lives on %s: the geobi path runs on the MI355X only (no CPU fallback)' % (what, t.device))


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_fn_cache = {}


def call(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(lib(), name)
    if fn(*args) != 0:
        check(1, name)


def read_i32(t, n=None):
    """Device int32 tensor -> list of Python ints through geobi_read_i32 (spinning wait on a pinned copy)."""
    n = t.numel() if n is None else n
    buf = (ctypes.c_int32 * n)()
    call('geobi_read_i32', ptr(t), n, buf, stream())
    return list(buf)


_size_cache = {}


def size_query(name, *args):
    """Memoised host-side size queries (geobi_*_ws_bytes, geobi_feast_ldz): pure functions of their
    integer arguments, called once per op otherwise (~3 us of ctypes each)."""
    key = (name,) + args
    v = _size_cache.get(key)
    if v is None:
        if len(_size_cache) > 8192:
            _size_cache.clear()
        v = _size_cache[key] = getattr(lib(), name)(*args)
    return v
