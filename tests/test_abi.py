"""The C-ABI boundary: libacm_hip.so loads without a GPU, exports every symbol the headers declare,
keeps the reference's struct layouts, and refuses to decode (loudly) when no HIP device exists."""
import ctypes as C
import os
import re

import pytest

import oracle_api as O
from helpers import golden_file
from libacm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(acm[a-z_]*_[a-z_0-9]+|acm_[a-z_]+)\s*\(", src))


def test_exports_every_declared_symbol():
    L = capi.lib()
    want = declared_functions("acm_hip.h") | declared_functions("libacm.h")
    want -= {"acm_io_callbacks"}
    assert set(capi.ACMHIP_SYMBOLS) | set(capi.LIBACM_SYMBOLS) == want, \
        sorted(want ^ (set(capi.ACMHIP_SYMBOLS) | set(capi.LIBACM_SYMBOLS)))
    for name in sorted(want):
        assert getattr(L, name) is not None, name
    assert len(capi.LIBACM_SYMBOLS) == 19          # reference libacm.h:120-170


def test_synth_library_exports():
    from libacm_amd import synth
    for name in ("acmsynth_defaults", "acmsynth_bound", "acmsynth_generate"):
        assert getattr(synth.lib(), name)


def test_struct_layouts_match_reference_abi():
    # ACMInfo 32 B, acm_io_callbacks 32 B; ACMStream 176 B is static_assert'ed in acm_stream.cpp
    assert C.sizeof(O.Info) == 32
    assert C.sizeof(O.IoCallbacks) == 32
    assert C.sizeof(capi.BlkHdr) == 8
    assert C.sizeof(capi.Patch) == 16
    assert C.sizeof(capi.StreamDesc) == 48
    # public fields callers of the reference read directly: info@0, total_values@32, data_len@80, block_len@120
    src = golden_file("f7_src")
    s = O.LibacmStream(O.bind_libacm(capi.lib()), src)
    raw = (C.c_uint8 * 176).from_address(s.h.value)
    words = memoryview(raw).cast("B").cast("I")
    info = s.info()
    assert words[0] == info["channels"] and words[1] == info["rate"] and words[5] == info["acm_level"]
    assert words[8] == 7 * 6 * 32 - 7                  # total_values
    assert words[20] == len(src)                        # data_len
    assert words[30] == info["acm_rows"] * info["acm_cols"]      # block_len
    assert words[31] == 2 * info["acm_cols"] - 2        # wrapbuf_len
    s.close()


def test_error_codes_and_strings():
    L = O.bind_libacm(capi.lib())
    want = {0: "No error", -1: "ACM error", -2: "Cannot open file", -3: "Not an ACM file", -4: "Read error",
            -5: "Bad format", -6: "Corrupt file", -7: "Unexcpected EOF", -8: "Stream not seekable",
            -9: "Unknown error", 1: "Unknown error", -100: "Unknown error"}
    for e, text in want.items():
        assert L.acm_strerror(e).decode() == text


@pytest.mark.skipif(capi.device_count() > 0, reason="a HIP device is present")
def test_no_device_decodes_on_the_host_and_device_calls_fail_loudly():
    """No GPU: everything that asks for the DEVICE (handles, plans, pinned memory, the batch call) fails loudly - those calls have no
    fallback.  acm_read() - the reference's own API, which decodes anywhere (decode.c:826-876) - synthesises on the host with the
    library's own host code (libacm_amd/csrc/acm_host_synth.cpp; never the test oracle), bit-exact with the goldens."""
    with pytest.raises(capi.AcmHipError):
        capi.Device(0)
    assert capi.lib().acmhip_last_error()
    src = golden_file("f7_src")
    s = O.LibacmStream(O.bind_libacm(capi.lib()), src)
    assert s.err == 0
    rc, _ = s.read(0, discard=True)
    assert s.read(64, discard=True)[0] == 64            # decode-and-discard only parses
    rc, b = s.read(64)
    o = O.Oracle(src)
    o.read(64, discard=True)
    assert rc == 64 and b == o.read(64)[1]              # the PCM behind what was discarded, from the host synthesis
    o.close()
    s.close()
    p = C.c_void_p()
    assert capi.lib().acmhip_host_alloc(64, C.byref(p)) == capi.ERR_NO_DEVICE


def test_oracle_is_not_linked_into_the_product():
    """the shipped library must not contain or reference the test oracle"""
    blob = open(capi.lib_path(), "rb").read()
    assert b"acmo_" not in blob and b"acm_oracle" not in blob and b"libacm_ref" not in blob
