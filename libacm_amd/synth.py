"""ctypes binding of the synthetic ACM writer (include/acm_synth.h)."""
import ctypes as C
import os

from . import _build

MIX_SPEECH, MIX_UNIFORM, MIX_SINGLE = 0, 1, 2
VALID_CODES = [0] + list(range(3, 17)) + [17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29]
BAD_CODES = [1, 2, 25, 28, 30, 31]
BASE_SEED = 0xAC3D0000  # seed = BASE_SEED + stream_id


class Params(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("level", C.c_uint32), ("rows", C.c_uint32),
                ("nblocks", C.c_uint32), ("channels", C.c_uint32), ("rate", C.c_uint32),
                ("total_values", C.c_uint32), ("pwr_min", C.c_uint32), ("pwr_max", C.c_uint32),
                ("val_min", C.c_uint32), ("val_max", C.c_uint32), ("mix", C.c_uint32),
                ("single_code", C.c_uint32), ("wavc", C.c_uint32), ("allow_out_of_range", C.c_uint32),
                ("prime_table", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build_synth()
        _lib = C.CDLL(path)
        _lib.acmsynth_defaults.argtypes = [C.POINTER(Params)]
        _lib.acmsynth_defaults.restype = None
        _lib.acmsynth_bound.argtypes = [C.POINTER(Params)]
        _lib.acmsynth_bound.restype = C.c_size_t
        _lib.acmsynth_generate.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t]
        _lib.acmsynth_generate.restype = C.c_size_t
    return _lib


def make_params(**kw):
    p = Params()
    lib().acmsynth_defaults(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown synth parameter %r" % k)
        setattr(p, k, v)
    return p


def generate(**kw):
    """Return the bytes of one synthetic ACM file.  Keywords = acmsynth_params fields."""
    p = make_params(**kw)
    cap = lib().acmsynth_bound(C.byref(p))
    buf = (C.c_uint8 * cap)()
    n = lib().acmsynth_generate(C.byref(p), buf, cap)
    if n == 0:
        raise ValueError("acmsynth_generate rejected the parameters")
    return bytes(buf[:n])


def generate_into(arr, **kw):
    """Write one file image into a writable uint8 numpy array; returns the byte count."""
    p = make_params(**kw)
    n = lib().acmsynth_generate(C.byref(p), arr.ctypes.data, arr.nbytes)
    if n == 0:
        raise ValueError("acmsynth_generate: buffer too small or bad parameters")
    return n


def bound(**kw):
    p = make_params(**kw)
    return lib().acmsynth_bound(C.byref(p))
