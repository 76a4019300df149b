"""ctypes binding of libsemseg_hip.so.  Signatures are parsed from include/semseg_hip.h (single source of truth).

The library is mandatory: there is NO fallback path.  ``lib()`` raises if the shared object is missing or a
declared symbol is not exported.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'semseg_hip.h')
LIB_PATH = os.environ.get('SEMSEG_HIP_LIB') or os.path.join(HERE, 'libsemseg_hip.so')     # the override serves kernel A/B builds (tools/)

_CTYPES = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'double': ctypes.c_double, 'hipStream_t': ctypes.c_void_p}
_lib = None


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(const char\*|int64_t|int)\s+(ss_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = {'const char*': ctypes.c_char_p, 'int64_t': ctypes.c_int64, 'int': ctypes.c_int}[ret]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace('const ', '').split()[0]
                    argtypes.append(_CTYPES[ty])
        protos[name] = (restype, argtypes)
    return protos


TRACE = None            # bench.py sets this to a list to time every entry point with HIP events on the launch stream


class _Traced:
    """Wraps the CDLL so that each ss_* call is bracketed by two HIP events recorded on torch's current stream (the stream
    every launch of this package uses) and logged as (name, args, start_event, end_event)."""

    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if name.endswith(('_workspace', '_ok')) or name in ('ss_last_error', 'ss_version'):     # host-only queries launch nothing
            return fn
        import torch

        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            TRACE.append((name, args, e0, e1))
            return rc
        return call


def lib():
    if TRACE is not None:
        return _Traced(_load())
    return _load()


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                               '(semseg_amd has no fallback path)')
        import torch                        # noqa: F401  torch's bundled HIP runtime must be the one already in the process:
        #                                     loading this library first would pull in /opt/rocm's second copy of libamdhip64
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(handle, name, None)
            if fn is None:
                raise RuntimeError(f'libsemseg_hip.so does not export {name} declared in include/semseg_hip.h')
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f'semseg_hip error {rc}: {lib().ss_last_error().decode()}')
