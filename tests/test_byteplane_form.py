"""The byte-plane staged form (include/acm_hip.h, libacm_amd/csrc/acm_pack.cpp) and the coefficient tables of the matrix-core first
pass (tools/gen_mfma_tables.py -> libacm_amd/csrc/acm_mfma_tables.inc), on the CPU.

The form is the int16 form's bytes in another order, so the round trip must be exact; the tables are checked against the CPU
oracle's own juggle cascade (reference decode.c:527-590): three stages over a residue class of the columns, computed as
A x inputs with the staged bytes exactly as the kernel's matrix instruction sees them, must equal what the oracle leaves in its
block buffer after stage 2 (sub_count 2, 4, 8)."""
import ctypes as C
import re

import numpy as np
import pytest

from helpers import make_stream
from libacm_amd import capi

LEVELS = [7, 8, 9, 10, 11, 12, 13, 14]


def test_levels_with_a_byteplane_form():
    L = capi.lib()
    for level in range(16):
        tr = L.acmhip_mform_tile_rows(level)
        if level in LEVELS:
            qn = L.acmhip_mform_group(level)
            assert qn in (8, 16, 64)
            # a level of the chunk kernel (64 columns of a residue class side by side) is cut into chunks of 2048 samples; levels 13 and 14
            # (the same first pass inside acm_tile2) into row pairs
            assert tr == ((2 if level >= 13 else max(1, 2048 >> level)) if qn == 64 else {12: 4, 13: 2, 14: 2}.get(level, 8192 >> level))
            assert L.acmhip_mform_bytes(level, 10) == 10 * (2 << level) + (2 << level) + 64
        else:
            assert tr == 0 and L.acmhip_mform_group(level) == 0
            assert L.acmhip_mform_rows(level, None, 0, None, 0, None, None) != 0


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("rows,pwr_max", [(1, 12), (16, 12), (16, 3), (17, 15)])
def test_round_trip(level, rows, pwr_max):
    tr = capi.lib().acmhip_mform_tile_rows(level)
    nblocks = (3 * max(tr, 4) + rows - 1) // rows + 1
    qn = capi.lib().acmhip_mform_group(level)            # columns of a residue class side by side: 8, 16 or 64
    split = qn == 64                                     # the chunk kernel's form: 8 and 16 bits only, idx = 256 hi + lo with both bytes signed
    s = capi.stage_file(make_stream(41000 + level * 100 + rows, level, rows, nblocks, pwr_min=min(2, pwr_max), pwr_max=pwr_max,
                                    val_max=65535 if pwr_max == 15 else 255))
    if split and level >= 13 and int(s.idx.max()) >= 32640:
        s.idx[s.idx >= 32640] = 32639                    # (beyond that the form of levels 13 / 14 has no place for an index: test_split_form_range)
    cols = 1 << level
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows * cols, level=level, rows=rows,
                        nrows=s.info.blocks * rows, row_begin=0)
    mf = capi.mform_streams(s.idx, [d])
    nt = mf.streams[0].ntiles
    assert nt >= 3 and mf.streams[0].form == capi.FORM_BYTEPLANE and mf.streams[0].chunk_off == 0
    nrows = nt * tr
    npairs = nrows // 2 + 1
    assert capi.lib().acmhip_mform_pairs(nrows) == npairs
    pairs = mf.pairs[:npairs]
    cls, off = pairs & 3, (pairs >> 2).astype(np.int64) * 64
    # the pair in front of the stream: index 0 everywhere, at 4 bits (nibble value 8) - at 8 bits in the chunk kernel's form
    if split:
        assert cls[0] == 2 and off[0] == 0 and (mf.data[:2 * cols] == 0).all()
    else:
        assert cls[0] == 1 and off[0] == 0 and (mf.data[:cols] == 0x88).all()
    # every pair at the narrowest class that holds it, one behind the other
    rowsv = s.idx[:nrows * cols].reshape(nrows // 2, 2 * cols).astype(np.int64)
    lo, hi = rowsv.min(axis=1), rowsv.max(axis=1)
    nib12 = split and level <= 12                        # the chunk kernel's own levels: a 12-bit class (signed low byte + signed high nibble)
    want_cls = np.where((lo >= -8) & (hi <= 7) & (not split), 1, np.where((lo >= -128) & (hi <= 127), 2,
                        np.where((lo >= -2176) & (hi <= 1919) & nib12, 1, np.where((hi >= 32640) & nib12, 0, 3))))
    assert np.array_equal(cls[1:], want_cls)
    size = np.array([4 * cols, 3 * cols if nib12 else cols, 2 * cols, 4 * cols])[cls]
    assert np.array_equal(off[1:], off[:-1] + size[:-1])
    assert off[-1] + size[-1] + 64 <= mf.data.size <= capi.lib().acmhip_mform_bytes(level, nrows) + 256
    back = capi.mform_unrows(level, mf.data, pairs, nrows)
    assert np.array_equal(back, s.idx[:nrows * cols])
    # layout: pair p, row r of it, residue c, q
    sigma = cols // qn
    rng = np.random.default_rng(level)
    for _ in range(96):
        p, r, c, q = int(rng.integers(nrows // 2)), int(rng.integers(2)), int(rng.integers(sigma)), int(rng.integers(qn))
        x = int(s.idx[(2 * p + r) * cols + c + q * sigma])
        k = int(cls[p + 1])
        at = int(off[p + 1]) + r * int(size[p + 1]) // 2
        if k == 0:
            # levels 8-12, a pair with an index beyond 32639: the low byte unsigned, stored minus 128; the arithmetic high byte
            assert split and mf.data[at + 2 * qn * c + q] == (x & 0xFF) ^ 0x80 and mf.data[at + 2 * qn * c + qn + q] == (x >> 8) & 0xFF
        elif k == 3 and split:
            lo_s = ((x & 0xFF) ^ 0x80) - 0x80
            assert mf.data[at + 2 * qn * c + q] == lo_s & 0xFF and mf.data[at + 2 * qn * c + qn + q] == ((x - lo_s) >> 8) & 0xFF
            assert -128 <= (x - lo_s) >> 8 <= 127
        elif k == 3:
            assert mf.data[at + 2 * qn * c + q] == (x & 0xFF) ^ 0x80 and mf.data[at + 2 * qn * c + qn + q] == (x >> 8) & 0xFF
        elif k == 2:
            assert mf.data[at + qn * c + q] == x & 0xFF
        elif split:
            # 12 bits: per residue 64 signed low bytes, then 32 bytes of signed high nibbles in lane order - lane ks (columns 16 ks .. + 15
            # of the class) reads 8 bytes at 8 ks: dword d holds its elements 8 d .. 8 d + 7, element 8 d + b in the HIGH nibble of byte b,
            # element 8 d + 4 + b in the low one
            lo_s = ((x & 0xFF) ^ 0x80) - 0x80
            h = (x - lo_s) >> 8
            assert -8 <= h <= 7 and mf.data[at + 96 * c + q] == lo_s & 0xFF
            ks, e = divmod(q, 16)
            dw, e8 = divmod(e, 8)
            byte = int(mf.data[at + 96 * c + 64 + 8 * ks + 4 * dw + (e8 % 4)])
            assert ((byte >> 4) if e8 < 4 else (byte & 15)) == h & 15
        else:
            j, i = divmod(q, 8)
            nib = 2 * (i % 4) + (i // 4)
            byte = mf.data[at + (qn // 2) * c + 4 * j + nib // 2]
            assert (byte >> (4 * (nib % 2))) & 15 == x + 8


def test_twelve_bit_class():
    """levels 8-12 (acm_chunk): a row pair whose indices fit a signed low byte + a signed high nibble, [-2176, 1919], takes 1.5 bytes per
    index; one index beyond either end and the pair is written at 16 bits.  Levels 13 / 14 (the same form read by acm_tile2) never get it."""
    L = capi.lib()
    for level in (8, 9, 10, 11, 12, 13):
        cols, tr = 1 << level, L.acmhip_mform_tile_rows(level)
        nrows = 2 * max(tr, 2) + 2
        rng = np.random.default_rng(1200 + level)
        for lo_v, hi_v, want in ((-2176, 1919, 1), (-2177, 0, 3), (0, 1920, 3), (-128, 127, 2), (-129, 0, 1), (0, 128, 1)):
            idx = rng.integers(max(lo_v, -2176), min(hi_v, 1919) + 1, size=nrows * cols).astype(np.int16)
            idx[2 * cols + 7], idx[3 * cols + cols - 3] = lo_v, hi_v         # the extremes sit in the second pair
            if want == 2:
                idx[:] = np.clip(idx, -128, 127)
            buf = np.zeros(L.acmhip_mform_bytes(level, nrows) + 256, dtype=np.uint8)
            pairs = np.zeros(nrows // 2 + 33, dtype=np.uint32)
            used = C.c_uint64()
            assert L.acmhip_mform_rows(level, idx.ctypes.data, nrows, buf.ctypes.data, 0, pairs.ctypes.data, C.byref(used)) == 0
            cls = pairs[1:nrows // 2 + 1] & 3
            if level >= 13:
                assert (cls >= 2).all()
            else:
                assert cls[1] == want, (level, lo_v, hi_v, cls)
                if want == 1:
                    nxt = (int(pairs[3]) >> 2) - (int(pairs[2]) >> 2)
                    assert nxt * 64 == 3 * cols
            assert np.array_equal(capi.mform_unrows(level, buf, pairs, nrows), idx)
    # a stream of pwr 8 ... 10 blocks: every pair at 12 bits, 1.5 bytes per index
    level, rows = 9, 16
    s = capi.stage_file(make_stream(47000, level, rows, 12, pwr_min=8, pwr_max=10))
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows << level, level=level, rows=rows, nrows=s.info.blocks * rows, row_begin=0)
    mf = capi.mform_streams(s.idx, [d])
    assert ((mf.pairs[1:1 + s.info.blocks * rows // 2] & 3) == 1).all()
    assert mf.data.size <= 1.5 * (s.info.blocks * rows << level) + (2 << level) + 64 + 256


def test_split_form_range():
    """the six-stage form writes a 16-bit index as two SIGNED bytes, 256 hi + lo: that ends at 32639.  A pair with a larger index is written
    in the whole-range class instead (code 0: the low byte unsigned, stored minus 128) and reads back exactly - at every level of the form
    (8-14: round 6 taught FirstPassZW of levels 13 / 14 the class); no stream is refused for its indices (ACMHIP_ERR_RANGE is history)"""
    L = capi.lib()
    levels = [lv for lv in LEVELS if L.acmhip_mform_group(lv) == 64]
    assert levels
    for level in levels:
        cols, tr = 1 << level, L.acmhip_mform_tile_rows(level)
        nrows = 2 * max(tr, 2)
        for top in (32639, 32640, 32767):
            idx = np.zeros(nrows * cols, dtype=np.int16)
            idx[5], idx[cols + 9], idx[3 * cols - 1] = top, -32768, -129
            buf = np.zeros(L.acmhip_mform_bytes(level, nrows) + 256, dtype=np.uint8)
            pairs = np.zeros(nrows // 2 + 33, dtype=np.uint32)
            used = C.c_uint64()
            rc = L.acmhip_mform_rows(level, idx.ctypes.data, nrows, buf.ctypes.data, 0, pairs.ctypes.data, C.byref(used))
            assert rc == 0, (level, top, rc)
            assert np.array_equal(capi.mform_unrows(level, buf, pairs, nrows), idx)
            assert int(pairs[1]) & 3 == (0 if top >= 32640 else 3)          # the first pair holds `top`
        d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=nrows * cols, level=level, rows=1, nrows=nrows, row_begin=0)
        assert capi.mform_streams(idx, [d]).streams[0].ntiles > 0           # (idx still holds 32767)


def test_width_classes_follow_the_blocks():
    """quiet blocks (pwr <= 3: indices in [-8, 7]) travel at 4 bits, pwr <= 7 at 8 bits: the stager's classes are what the block
    headers promise or narrower"""
    level, rows = 7, 16             # (the level of acm_tile2's three-stage matrix build: the six-stage form has no 4-bit class)
    s = capi.stage_file(make_stream(44000, level, rows, 16, pwr_min=0, pwr_max=12))
    cols = 1 << level
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows * cols, level=level, rows=rows,
                        nrows=s.info.blocks * rows, row_begin=0)
    mf = capi.mform_streams(s.idx, [d])
    cls = (mf.pairs[1:1 + s.info.blocks * rows // 2] & 3).reshape(s.info.blocks, rows // 2)
    pwr = s.hdr[:s.info.blocks, 1]
    assert len(set(pwr.tolist())) > 4
    for b in range(s.info.blocks):
        bound = 1 if pwr[b] <= 3 else 2 if pwr[b] <= 7 else 3
        assert cls[b].max() <= bound, (b, pwr[b], cls[b])
    assert (cls == 1).any() and (cls == 2).any() and (cls == 3).any()
    assert mf.nbytes < 0.9 * s.info.blocks * rows * cols * 2


def load_tables(G):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    txt = open(os.path.join(here, "..", "libacm_amd", "csrc", "acm_mfma_tables.inc")).read()
    out = {}
    B = 2 << G
    for name, shape in (("A", (2, B, 2 * B)), ("KROW", (2, 4, B)), ("BIAS", (2, 2, B))):
        body = txt[txt.index("ACM_MF_%s%d[" % (name, G)):]
        body = body[body.index("=") + 1:body.index(";")]
        out[name] = np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int64).reshape(shape)
    return out


def juggle_stages(x, cols, nstages):
    """the first nstages stages of juggle_block over rows x cols values (reference decode.c:527-577), written as the strided 3-tap FIR
    each stage is: over the row-major sequence, with stride s = the stage's sub_len,
    y[m] = 2 x[m - s] + sigma (x[m - 2 s] + x[m]), sigma = -1 where m // s is odd, zeros in front, "+1" on m % s == 0 after stage 0
    (decode.c:561-564).  test_fir_restatement_matches_the_oracle pins this to the oracle."""
    rows = x.shape[0]
    flat = x.astype(np.int64).reshape(-1)
    m = np.arange(flat.size)
    s = cols // 2
    for st in range(nstages):
        x1 = np.where(m >= s, flat[np.maximum(m - s, 0)], 0)
        x2 = np.where(m >= 2 * s, flat[np.maximum(m - 2 * s, 0)], 0)
        flat = 2 * x1 + np.where((m // s) & 1, -1, 1) * (x2 + flat)
        if st == 0:
            flat = flat + (m % s == 0)
        s //= 2
    return flat.reshape(rows, cols)


def test_fir_restatement_matches_the_oracle():
    """the stage formula used below is the oracle's: all `level` stages of it reproduce the oracle's PCM of a small stream"""
    import oracle_api as O
    level, rows = 7, 5
    f = make_stream(42000, level, rows, 3, pwr_max=9)
    s = capi.stage_file(f)
    cols = 1 << level
    nrows = s.info.blocks * rows
    val = np.repeat(s.hdr[:s.info.blocks, 0].astype(np.int64), rows)
    x = s.idx[:nrows * cols].astype(np.int64).reshape(nrows, cols) * val[:, None]
    y = juggle_stages(x, cols, level)
    want = O.Oracle.decode_all(f)[0].astype(np.int64)
    got = ((y.reshape(-1) & 0xFFFFFFFF) >> level).astype(np.uint16).astype(np.int16).astype(np.int64)
    assert np.array_equal(got[:want.size], want)


@pytest.mark.parametrize("G", [3, 4])
@pytest.mark.parametrize("level", [8, 10])
def test_matrix_tables_reproduce_the_first_stages(level, G):
    """A x (index bytes) + KROW, times val, + BIAS == output of stage G - 1 of the cascade, for every residue class and row pair of a
    staged stream (one val over all rows here; the per-row split is the same sum taken row by row).  The bytes are what the kernel's
    matrix instruction reads: low byte minus 128 and high byte, both signed."""
    t = load_tables(G)
    A, KROW, BIAS = t["A"][0], t["KROW"][0], t["BIAS"][0]
    qn = 1 << G
    cols, sigma = 1 << level, (1 << level) // qn
    rows = 6
    s = capi.stage_file(make_stream(43000 + level, level, rows, 1, pwr_max=15, val_max=65535))
    assert s.info.blocks == 1
    val = int(s.hdr[0, 0])
    idx = s.idx[:rows * cols].astype(np.int64).reshape(rows, cols)
    want = juggle_stages(idx * val, cols, G)
    padded = np.concatenate([np.zeros((2, cols), dtype=np.int64), idx])          # two rows of index 0 in front of the stream
    lo = ((padded & 0xFF) ^ 0x80).astype(np.uint8).view(np.int8).astype(np.int64)
    hi = (padded >> 8).astype(np.int64)
    assert np.array_equal(lo + 128 + 256 * hi, padded)
    for p in range(rows // 2):
        for c in range(sigma):
            blo = lo[2 * p:2 * p + 4, c::sigma].reshape(4 * qn)                    # rows 2p-2 .. 2p+1, the residue's qn columns each
            bhi = hi[2 * p:2 * p + 4, c::sigma].reshape(4 * qn)
            d = A @ blo + KROW.sum(axis=0) + ((A @ bhi) << 8)
            assert np.abs(d).max() < 1 << 23                                       # the kernel multiplies with the 24-bit multiplier
            y = d * val + (BIAS[1 if p == 0 else 0] if c == 0 else 0)
            got = y.reshape(2, qn)                                                 # output m = qn * (row in pair) + q
            w = want[2 * p:2 * p + 2, c::sigma]
            assert np.array_equal(got, w), (p, c)
    # variant 1 = the same with odd outputs negated (what a P stage leaves in LDS)
    sign = np.where(np.arange(2 * qn) & 1, -1, 1)
    assert np.array_equal(t["A"][1], A * sign[:, None])
    assert np.array_equal(t["KROW"][1], KROW * sign[None, :]) and np.array_equal(t["BIAS"][1], BIAS * sign[None, :])


def load_toeplitz(G=6):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    txt = open(os.path.join(here, "..", "libacm_amd", "csrc", "acm_toeplitz_tables.inc")).read()
    U = 1 << G
    out = {}
    for name, key, shape in (("T", "ACM_TZ%d[" % G, (2, 3, U, U)), ("BIAS", "ACM_TZ%d_BIAS[" % G, (2, 3, U))):
        body = txt[txt.index(key):]
        body = body[body.index("=") + 1:body.index(";")]
        out[name] = np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int64).reshape(shape)
    return out


@pytest.mark.parametrize("level", [9, 10])
def test_toeplitz_tables_reproduce_the_first_six_stages(level):
    """acm_chunk's first pass: over one residue class (columns c + q cols/64) the first six stages of the cascade are
    out[r] = T0 x[r] + T1 x[r-1] + T2 x[r-2] on the rows' 64 indices each; with the index split into two SIGNED bytes (the staged form of the
    kernel's levels), the products scaled by val and the "+1" response added for the rows that exist, that is what the oracle's stage formula
    leaves after stage 5 - for every class and row of a staged stream, the first rows of the stream included.  The second storage convention
    is the first with the odd outputs negated."""
    t = load_toeplitz()
    T, BIAS = t["T"][0], t["BIAS"][0]
    assert np.abs(t["T"]).max() <= 64
    cols, sigma, rows = 1 << level, (1 << level) // 64, 7
    s = capi.stage_file(make_stream(43500 + level, level, rows, 1, pwr_max=15, val_max=65535))
    val = int(s.hdr[0, 0])
    idx = np.minimum(s.idx[:rows * cols].astype(np.int64), 32639).reshape(rows, cols)
    want = juggle_stages(idx * val, cols, 6)
    lo = ((idx & 0xFF) ^ 0x80) - 0x80
    hi = (idx - lo) >> 8
    assert hi.min() >= -128 and hi.max() <= 127 and np.array_equal(256 * hi + lo, idx)
    z = np.zeros(64, dtype=np.int64)
    for r in range(rows):
        for c in range(sigma):
            x = [(lo[r - j, c::sigma], hi[r - j, c::sigma]) if r - j >= 0 else (z, z) for j in range(3)]
            dl = sum(T[j] @ x[j][0] for j in range(3))
            dh = sum(T[j] @ x[j][1] for j in range(3))
            assert max(np.abs(dl).max(), np.abs(dh).max()) < 1 << 23                  # the kernel multiplies with the 24-bit multiplier
            y = (dl + (dh << 8)) * val + (BIAS[min(r, 2)] if c == 0 else 0)
            assert np.array_equal(y, want[r, c::sigma]), (r, c)
    sign = np.where(np.arange(64) & 1, -1, 1)
    assert np.array_equal(t["T"][1], T * sign[None, :, None]) and np.array_equal(t["BIAS"][1], BIAS * sign[None, :])


@pytest.mark.parametrize("level,rows,blocks,cut", [(9, 16, 7, 0), (9, 16, 5, 3), (7, 16, 20, 0), (10, 8, 11, 5), (11, 64, 2, 0), (12, 4, 9, 1), (8, 2, 70, 0),
                                                    (9, 17, 9, 0), (9, 3, 43, 2), (8, 1, 140, 0), (10, 33, 5, 7), (7, 5, 60, 1), (11, 1, 37, 0), (12, 3, 11, 0)])
def test_fused_staging_equals_the_two_pass_staging(level, rows, blocks, cut):
    """acm_stage_file_mform (the byte-plane form written block by block by the parsing pass) leaves exactly what acm_stage_file +
    acmhip_mform_rows leave: the same pair table and bytes for the whole tiles, the same int16 rows from two rows in front of the ragged
    tail on - and touches no int16 row in front of that"""
    L = capi.lib()
    f = make_stream(46000 + level * 10 + rows, level, rows, blocks, cut=cut, pwr_max=12)
    s = capi.stage_file(f)
    info, idx, hdr, blob, pairs, mf_rows, mf_bytes = capi.stage_file_mform(f, mf_base=4096)
    cols = 1 << level
    t2 = L.acmk_tile2_rows(level)
    words = min(s.info.blocks * rows * cols, s.info.total_values)
    want_rows = min(s.info.blocks * rows, words >> level) // t2 * t2
    assert mf_rows == want_rows and mf_rows > 0 and info.blocks == s.info.blocks and info.end_status == s.info.end_status
    assert np.array_equal(hdr[:info.blocks], s.hdr)
    # the two-pass form of the same rows
    buf = np.zeros(L.acmhip_mform_bytes(level, mf_rows) + 256, dtype=np.uint8)
    pr = np.zeros(mf_rows // 2 + 33, dtype=np.uint32)
    used = C.c_uint64()
    assert L.acmhip_mform_rows(level, s.idx.ctypes.data, mf_rows, buf.ctypes.data, 4096, pr.ctypes.data, C.byref(used)) == 0
    assert used.value == mf_bytes and np.array_equal(pairs[:mf_rows // 2 + 1], pr[:mf_rows // 2 + 1])
    assert np.array_equal(blob[:mf_bytes], buf[:mf_bytes])
    tail = max(mf_rows - 2, 0) * cols
    n = info.blocks * rows * cols
    assert np.array_equal(idx[tail:n], s.idx[tail:n]) and (idx[:tail] == -12345).all()


def test_fused_staging_falls_back():
    """no form for: a level without one, H1 patches (npatches says so), a file that ends early, levels 13 / 14 - each time idx holds every row as acm_stage_file leaves it.  (Odd block heights - row pairs
    that straddle blocks - were on this list until round 6: test_fused_staging_equals_the_two_pass_staging has them now)"""
    cases = [make_stream(46500, 5, 16, 9), make_stream(46501, 4, 3, 40),
             make_stream(46502, 9, 16, 12, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6),
             make_stream(46504, 9, 16, 9)[:9000], make_stream(46505, 13, 4, 6)]
    for k, f in enumerate(cases):
        s = capi.stage_file(f)
        info, idx, hdr, blob, pairs, mf_rows, mf_bytes = capi.stage_file_mform(f)
        assert mf_rows == 0 and mf_bytes == 0, k
        assert info.blocks == s.info.blocks and info.end_status == s.info.end_status and info.npatches == s.info.npatches, k
        n = info.blocks * info.rows * info.cols
        assert np.array_equal(idx[:n], s.idx[:n]), k
    assert capi.stage_file(cases[2]).info.npatches > 0
    # (an index beyond 32639 kept a level-9 stream out of the form until round 6: now its pairs are written in the whole-range class)
    wide = make_stream(46503, 9, 16, 8, mix=2, single_code=16, pwr_min=15, pwr_max=15, val_max=65535)
    assert int(capi.stage_file(wide).idx.max()) >= 32640 and capi.stage_file_mform(wide)[5] > 0


def test_stager_rejects_what_the_kernel_could_not_read():
    """odd row counts (a unit is a row pair), a block that would not start on 16 bytes, offsets beyond the pair table's 30 bits, levels
    without the form; the inverse refuses a table with an unknown width class or a non-zero pair in front"""
    L = capi.lib()
    level, cols = 7, 128             # (the form with a 4-bit class)
    idx = np.zeros(4 * cols, dtype=np.int16)
    out = np.zeros(L.acmhip_mform_bytes(level, 4), dtype=np.uint8)
    pairs = np.zeros(8, dtype=np.uint32)
    used = C.c_uint64()
    args = lambda lv, n, base: (lv, idx.ctypes.data, n, out.ctypes.data, base, pairs.ctypes.data, C.byref(used))
    assert L.acmhip_mform_rows(*args(level, 4, 0)) == 0 and used.value == cols + 2 * cols + 64         # three pairs at 4 bits + slack
    assert (pairs[:3] & 3 == 1).all() and pairs[3] == 0
    assert L.acmhip_mform_rows(*args(level, 3, 0)) != 0
    assert L.acmhip_mform_rows(*args(level, 4, 16)) != 0
    assert L.acmhip_mform_rows(*args(level, 4, 1 << 36)) != 0
    assert L.acmhip_mform_rows(*args(level, 4, 1 << 34)) == 0 and pairs[0] >> 2 == 1 << 28
    assert L.acmhip_mform_rows(*args(5, 4, 0)) != 0
    assert L.acmhip_mform_rows(*args(level, 4, 0)) == 0
    back = np.zeros(4 * cols, dtype=np.int16)
    assert L.acmhip_mform_unrows(level, out.ctypes.data, pairs.ctypes.data, 4, back.ctypes.data) == 0
    bad = pairs.copy()
    bad[2] &= ~np.uint32(3)
    assert L.acmhip_mform_unrows(level, out.ctypes.data, bad.ctypes.data, 4, back.ctypes.data) != 0
    out[5] = 0x98                                   # an index of 1 in the pair in front of the stream
    assert L.acmhip_mform_unrows(level, out.ctypes.data, pairs.ctypes.data, 4, back.ctypes.data) != 0


def test_block_ranges_end_on_whole_tiles():
    """acmk_range_bound (acm_parse.hip; host and device share it): the block ranges of a device-parsed stream are cut at multiples of
    T / gcd(rows, T) blocks - the fewest blocks that are whole tiles of T rows - so that every range is a window on a tile boundary and no
    range cuts a row pair of an odd block height; the ranges tile the stream, short streams have empty ranges, unit 0 / 1 cuts anywhere"""
    from math import gcd
    L = capi.lib()
    L.acmk_range_bound.restype = C.c_uint32
    L.acmk_range_bound.argtypes = [C.c_uint32] * 4
    for blocks in (1, 2, 7, 16, 100, 251, 4001):
        for R in (1, 2, 3, 5, 16, 64):
            for rows, T in ((16, 16), (1, 16), (3, 32), (5, 8), (6, 4), (17, 2), (700, 16), (33, 4)):
                unit = T // gcd(rows, T)
                b = [L.acmk_range_bound(blocks, r, R, unit) for r in range(R + 1)]
                assert b[0] == 0 and b[R] == blocks and b == sorted(b), (blocks, R, unit, b)
                assert all(x % unit == 0 and (x * rows) % T == 0 for x in b[:-1])
                assert all(x <= blocks * r // R for r, x in enumerate(b[:-1]))          # never beyond the even cut: a range needs no more of the file than before
                plain = [L.acmk_range_bound(blocks, r, R, 1) for r in range(R + 1)]
                assert plain == [blocks * r // R for r in range(R)] + [blocks]
                assert plain == [L.acmk_range_bound(blocks, r, R, 0) for r in range(R + 1)]
