"""The product's HOST synthesis (libacm_amd/csrc/acm_host_synth.cpp, include/acm_hip.h acmhip_host_synth) against the CPU oracle.

It restates, in the kernels' own formulation (strided three-tap stages over the flat sample index, history = the two staged rows in
front), what the reference does in decode.c:586-600 (unpack), :508-577 (juggle / juggle_block) and :617-677 (the four writers).  The
oracle (oracle/acm_oracle.c, pinned to the compiled reference and the goldens) is only the checker here; nothing of it is linked into
the library (tests/test_abi.py).  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from helpers import fmt_args, make_stream, oracle_pcm
from libacm_amd import capi


def host_synth(staged, fmt=capi.FMT_S16LE, row_begin=0, n_emit=None, patches=True):
    info = staged.info
    cols = 1 << info.level
    nrows = info.blocks * info.rows
    if n_emit is None:
        n_emit = min(info.total_values, nrows * cols) - row_begin * cols
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=n_emit, level=info.level, rows=info.rows, nrows=nrows, row_begin=row_begin)
    out = np.full(n_emit + 8, 0xA5A5, dtype=np.uint16)
    pl = list(staged.patches) if (patches and staged.patches is not None) else []
    arr = (capi.Patch * max(1, len(pl)))(*pl)
    rc = capi.lib().acmhip_host_synth(C.byref(d), staged.idx.ctypes.data, staged.hdr.ctypes.data, arr if pl else None, len(pl), fmt, out.ctypes.data)
    assert rc == 0, rc
    assert (out[n_emit:] == 0xA5A5).all()               # nothing written behind the last sample asked for
    return out[:n_emit]


@pytest.mark.parametrize("level", list(range(16)))
def test_every_level_and_block_height(level):
    """levels 0-15 x block heights that put block boundaries everywhere x ragged ends: bit-exact PCM"""
    for rows in (1, 3, 16, 17) if level <= 11 else (1, 2, 5):
        nb = max(3, min(40, (1 << 16) // (rows << level)))
        f = make_stream(5000 + 17 * level + rows, level, rows, nb, cut=7 if level else 1)
        s = capi.stage_file(f)
        want, st = oracle_pcm(f)
        got = host_synth(s)
        assert got.size == want.size and np.array_equal(got, want), (level, rows)


@pytest.mark.parametrize("fmt", [capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE])
def test_output_formats(fmt):
    """the four writers of decode.c:617-655, incl. samples that wrap (no saturation) and 16-bit row values"""
    for lv, rows in ((7, 16), (9, 5), (2, 3)):
        f = make_stream(5400 + lv, lv, rows, 12, cut=3, val_max=65535, pwr_max=15)
        be, sg = fmt_args(fmt)
        want, _ = oracle_pcm(f, 0, be, sg)
        assert np.array_equal(host_synth(capi.stage_file(f), fmt=fmt), want)


def test_windows_need_only_the_two_rows_in_front():
    """a window that starts at any row (acm_read's read-ahead windows, seeks) equals the same samples of the whole decode"""
    for lv, rows, nb in ((7, 16, 9), (5, 1, 200), (10, 3, 6), (3, 7, 50)):
        f = make_stream(5500 + lv, lv, rows, nb)
        s = capi.stage_file(f)
        want, _ = oracle_pcm(f)
        cols = 1 << lv
        total_rows = nb * rows
        for rb in (1, 2, 3, rows, rows + 1, total_rows // 2, total_rows - 1):
            n = min(want.size - rb * cols, 5 * cols + 3)
            got = host_synth(s, row_begin=rb, n_emit=n)
            assert np.array_equal(got, want[rb * cols: rb * cols + n]), (lv, rows, rb)


def test_stale_table_patches():
    """hazard H1: indices outside the block's amplitude table are resolved by the host parser and shipped as patches (include/acm_hip.h);
    the host synthesis applies them like the kernels do - and without them the PCM differs (the test would otherwise prove nothing)"""
    seen = 0
    for seed in range(12):
        f = make_stream(5600 + seed, 6, 8, 30, pwr_min=0, pwr_max=3, mix=1)
        s = capi.stage_file(f)
        if s.patches is None or len(s.patches) == 0:
            continue
        seen += 1
        want, _ = oracle_pcm(f)
        assert np.array_equal(host_synth(s), want)
        assert not np.array_equal(host_synth(s, patches=False), want)
    assert seen >= 3


def test_argument_checks():
    L = capi.lib()
    f = make_stream(5700, 5, 4, 6)
    s = capi.stage_file(f)
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=5 * 4 * 32 + 1, level=5, rows=4, nrows=5 * 4, row_begin=0)
    out = np.zeros(2048, dtype=np.uint16)
    assert L.acmhip_host_synth(C.byref(d), s.idx.ctypes.data, s.hdr.ctypes.data, None, 0, 0, out.ctypes.data) == capi.ERR_ARG      # more samples than staged rows
    d.n_emit = 64
    assert L.acmhip_host_synth(C.byref(d), s.idx.ctypes.data, s.hdr.ctypes.data, None, 0, 7, out.ctypes.data) == capi.ERR_ARG      # no such format
    assert L.acmhip_host_synth(C.byref(d), s.idx.ctypes.data, s.hdr.ctypes.data, None, 0, 0, out.ctypes.data) == 0
    assert L.acmhip_host_synth_limit() == 1 << 27       # the default: streams below 128 Msamples stay on the host while no device is open


def test_acm_read_takes_the_host_path_for_short_streams_by_default():
    """the drop-in API with the library's defaults: a short stream is decoded by acm_read() without any device call - here (no GPU in the
    authoring container) and on a GPU box alike, as long as nothing has opened the default device"""
    import oracle_api as O
    for lv, rows, nb, ch in ((7, 16, 40, 1), (9, 16, 12, 2), (0, 3, 9, 1), (13, 2, 2, 1)):
        f = make_stream(5800 + lv, lv, rows, nb, channels=ch, cut=5)
        s = O.LibacmStream(O.bind_libacm(capi.lib()), f)
        pcm, rc = s.decode_all()
        s.close()
        want, _ = O.Oracle.decode_all(f)
        assert pcm == want.tobytes()
