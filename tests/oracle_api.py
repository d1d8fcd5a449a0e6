"""ctypes bindings of the TEST ORACLES (never imported by the product):

  Oracle  - our CPU restatement, oracle/_build/libacm_oracle.so
  Ref     - the real reference compiled from /root/reference, oracle/_ref/libacm_ref.so
            (present only where `make -C oracle ref` could run; it travels to the
            GPU box as a prebuilt file, its sources never do)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libacm_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libacm_ref.so")
REFPROBE_SO = os.path.join(ORACLE_DIR, "_ref", "libacm_refprobe.so")
REF_TOOL = os.path.join(ORACLE_DIR, "_ref", "acmtool_ref")

CLEAN_EOF = -99


def _ensure_built():
    src = [os.path.join(ORACLE_DIR, f) for f in ("acm_oracle.c", "acm_oracle.h")]
    if (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in src):
        subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(REF_SO):
        subprocess.run(["make", "-C", ORACLE_DIR, "ref"], check=True, stdout=subprocess.DEVNULL)


def have_ref():
    _ensure_built()
    return os.path.exists(REF_SO)


class Info(C.Structure):
    _fields_ = [(n, C.c_uint) for n in ("channels", "rate", "acm_id", "acm_version",
                                        "acm_channels", "acm_level", "acm_cols", "acm_rows")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# --------------------------------------------------------------------------
class Oracle:
    """Handle on one oracle stream over an in-memory file image."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            _ensure_built()
            L = C.CDLL(ORACLE_SO)
            L.acmo_open_mem.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_int]
            L.acmo_close.argtypes = [C.c_void_p]
            L.acmo_close.restype = None
            for f in ("acmo_read", "acmo_read_loop"):
                getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]
            L.acmo_seek_pcm.argtypes = [C.c_void_p, C.c_uint]
            L.acmo_seek_time.argtypes = [C.c_void_p, C.c_uint]
            L.acmo_get_info.argtypes = [C.c_void_p]
            L.acmo_get_info.restype = C.POINTER(Info)
            for f in ("acmo_bitrate", "acmo_rate", "acmo_channels", "acmo_raw_total", "acmo_raw_tell",
                      "acmo_pcm_total", "acmo_pcm_tell", "acmo_time_total", "acmo_time_tell",
                      "acmo_total_values", "acmo_block_len"):
                getattr(L, f).argtypes = [C.c_void_p]
                getattr(L, f).restype = C.c_uint
            L.acmo_seekable.argtypes = [C.c_void_p]
            L.acmo_strerror.argtypes = [C.c_int]
            L.acmo_strerror.restype = C.c_char_p
            L.acmo_fill_next_block.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.acmo_juggle_block.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
            L.acmo_juggle_block.restype = None
            L.acmo_output.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
            L.acmo_decode_all.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_uint,
                                          C.c_int, C.c_int, C.POINTER(C.c_int)]
            L.acmo_decode_all.restype = C.c_long
            L.acmo_decode_discard.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]
            L.acmo_decode_discard.restype = C.c_long
            cls._lib = L
        return cls._lib

    def __init__(self, data, force_chans=0, max_read=0, seekable=True):
        self._data = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        self.h = C.c_void_p()
        self.err = self.lib().acmo_open_mem(C.byref(self.h), self._data.ctypes.data, self._data.size,
                                            force_chans, max_read, 1 if seekable else 0)
        if self.err < 0:
            self.h = None

    def close(self):
        if self.h:
            self.lib().acmo_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read(self, nbytes, be=0, wordlen=2, sgned=1, discard=False, loop=False):
        fn = self.lib().acmo_read_loop if loop else self.lib().acmo_read
        if discard:
            return fn(self.h, None, nbytes, be, wordlen, sgned), b""
        buf = (C.c_uint8 * max(nbytes, 1))()
        rc = fn(self.h, buf, nbytes, be, wordlen, sgned)
        return rc, bytes(buf[:max(rc, 0)])

    def seek_pcm(self, pos):
        return self.lib().acmo_seek_pcm(self.h, pos)

    def seek_time(self, ms):
        return self.lib().acmo_seek_time(self.h, ms)

    def info(self):
        return self.lib().acmo_get_info(self.h).contents.as_dict()

    def getter(self, name):
        return getattr(self.lib(), "acmo_" + name)(self.h)

    def fill_next_block(self):
        n = self.getter("block_len")
        raw = np.empty(n, dtype=np.int32)
        pwr, val = C.c_int(), C.c_int()
        rc = self.lib().acmo_fill_next_block(self.h, raw.ctypes.data, C.byref(pwr), C.byref(val))
        return rc, raw, pwr.value, val.value

    # ---- stateless helpers ----
    @classmethod
    def decode_all(cls, data, force_chans=0, step_bytes=8192, be=0, sgned=1, cap_words=None):
        """-> (pcm int16 array in the requested byte layout viewed as native int16, status)"""
        a = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        if cap_words is None:
            o = cls(data, force_chans)
            if o.err < 0:
                return np.zeros(0, np.int16), o.err
            cap_words = o.getter("total_values")
            o.close()
        pcm = np.zeros(cap_words, dtype=np.int16)
        st = C.c_int()
        n = cls.lib().acmo_decode_all(a.ctypes.data, a.size, force_chans, pcm.ctypes.data, cap_words,
                                      step_bytes, be, sgned, C.byref(st))
        return pcm[:n], st.value

    @classmethod
    def decode_discard(cls, data, force_chans=0):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        st = C.c_int()
        n = cls.lib().acmo_decode_discard(a.ctypes.data, a.size, force_chans, C.byref(st))
        return n, st.value

    @classmethod
    def juggle_block(cls, level, rows, block, wrap):
        assert block.dtype == np.int32 and wrap.dtype == np.int32
        cls.lib().acmo_juggle_block(level, rows, block.ctypes.data, wrap.ctypes.data)

    @classmethod
    def output(cls, src, level, be, sgned, wordlen=2):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.zeros(src.size * 2, dtype=np.uint8)
        rc = cls.lib().acmo_output(src.ctypes.data, dst.ctypes.data, src.size, level, be, wordlen, sgned)
        return rc, dst

    @classmethod
    def strerror(cls, err):
        return cls.lib().acmo_strerror(err).decode()


# --------------------------------------------------------------------------
# The real reference, through its own public API (src/libacm.h) with memory callbacks.
READ_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
SEEK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
CLOSE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
LEN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class IoCallbacks(C.Structure):
    _fields_ = [("read_func", READ_FN), ("seek_func", SEEK_FN), ("close_func", CLOSE_FN),
                ("get_length_func", LEN_FN)]


class MemIO:
    """Python-side acm_io_callbacks over a bytes object (short reads, failures and
    missing callbacks are configurable) - usable with ANY library exporting libacm.h."""

    def __init__(self, data, max_read=0, seekable=True, with_length=True, fail_read_at=None):
        self.data = bytes(data)
        self.pos = 0
        self.max_read = max_read
        self.closed = 0
        self.read_calls = []
        self.fail_read_at = fail_read_at

        def _read(ptr, size, n, arg):
            want = size * n
            self.read_calls.append((size, n))
            if self.fail_read_at is not None and self.pos >= self.fail_read_at:
                return -1
            if self.max_read:
                want = min(want, self.max_read)
            chunk = self.data[self.pos:self.pos + want]
            C.memmove(ptr, chunk, len(chunk))
            self.pos += len(chunk)
            return len(chunk) // size if size else 0

        def _seek(arg, off, whence):
            if whence == 0:
                self.pos = off
            elif whence == 1:
                self.pos += off
            else:
                self.pos = len(self.data) + off
            return 0

        def _close(arg):
            self.closed += 1
            return 0

        def _len(arg):
            return len(self.data)

        self.cb = IoCallbacks()
        self._keep = (READ_FN(_read), SEEK_FN(_seek), CLOSE_FN(_close), LEN_FN(_len))
        self.cb.read_func = self._keep[0]
        if seekable:
            self.cb.seek_func = self._keep[1]
        self.cb.close_func = self._keep[2]
        if with_length:
            self.cb.get_length_func = self._keep[3]


def bind_libacm(L):
    """Attach libacm.h prototypes (src/libacm.h:120-170) to a loaded library."""
    L.acm_open_decoder.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, IoCallbacks, C.c_int]
    L.acm_open_file.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    for f in ("acm_read", "acm_read_loop"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int]
    L.acm_close.argtypes = [C.c_void_p]
    L.acm_close.restype = None
    L.acm_info.argtypes = [C.c_void_p]
    L.acm_info.restype = C.POINTER(Info)
    L.acm_seekable.argtypes = [C.c_void_p]
    for f in ("acm_bitrate", "acm_rate", "acm_channels", "acm_raw_total", "acm_raw_tell",
              "acm_pcm_total", "acm_pcm_tell", "acm_time_total", "acm_time_tell"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_uint
    L.acm_seek_pcm.argtypes = [C.c_void_p, C.c_uint]
    L.acm_seek_time.argtypes = [C.c_void_p, C.c_uint]
    L.acm_strerror.argtypes = [C.c_int]
    L.acm_strerror.restype = C.c_char_p
    return L


class LibacmStream:
    """One stream opened through the libacm.h API of `lib` (reference OR our drop-in)."""

    def __init__(self, lib, data, force_chans=0, **io_kw):
        self.L = lib
        self.io = MemIO(data, **io_kw)
        self.h = C.c_void_p()
        self.err = lib.acm_open_decoder(C.byref(self.h), None, self.io.cb, force_chans)
        if self.err < 0:
            self.h = None

    def close(self):
        if self.h:
            self.L.acm_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read(self, nbytes, be=0, wordlen=2, sgned=1, discard=False, loop=False):
        fn = self.L.acm_read_loop if loop else self.L.acm_read
        if discard:
            return fn(self.h, None, nbytes, be, wordlen, sgned), b""
        buf = (C.c_uint8 * max(nbytes, 1))()
        rc = fn(self.h, buf, nbytes, be, wordlen, sgned)
        return rc, bytes(buf[:max(rc, 0)])

    def seek_pcm(self, pos):
        return self.L.acm_seek_pcm(self.h, pos)

    def seek_time(self, ms):
        return self.L.acm_seek_time(self.h, ms)

    def info(self):
        return self.L.acm_info(self.h).contents.as_dict()

    def getter(self, name):
        return getattr(self.L, "acm_" + name)(self.h)

    def decode_all(self, step_bytes=8192, be=0, sgned=1):
        out = []
        while True:
            rc, b = self.read(step_bytes, be=be, sgned=sgned, loop=True)
            if rc <= 0:
                return b"".join(out), rc
            out.append(b)


_ref = None
_refprobe = None


def ref_lib():
    global _ref
    if _ref is None:
        _ensure_built()
        _ref = bind_libacm(C.CDLL(REF_SO))
    return _ref


def refprobe_lib():
    global _refprobe
    if _refprobe is None:
        _ensure_built()
        L = C.CDLL(REFPROBE_SO)
        L.refprobe_juggle_block.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.refprobe_juggle_block.restype = None
        L.refprobe_output.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _refprobe = L
    return _refprobe
