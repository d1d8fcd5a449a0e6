"""The packed staged form (include/acm_hip.h, libacm_amd/csrc/acm_pack.cpp): host packer against its inverse, on the CPU.

What is stored is what set_pos() would look up (reference decode.c:174-177): the filler indices, here per column pair and row
group at the narrowest of 0 / 4 / 8 / 16 bits (the ranges the fillers of decode.c:181-476 can produce).  Bit-exact round trip
is the bar; the GPU side of the form (acm_tile2p) is checked against the oracle in test_gpu_packed.py.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import make_stream
from libacm_amd import capi

LEVELS = [6, 7, 8, 9]


def staged_tiles(level, rows, ntiles, seed, **kw):
    tr = capi.packed_tile_rows(level)
    nblocks = (ntiles * tr + rows - 1) // rows + 1
    s = capi.stage_file(make_stream(seed, level, rows, nblocks, **kw))
    return s, tr


def test_levels_with_a_packed_form():
    L = capi.lib()
    for level in range(16):
        tr, gr = L.acmhip_packed_tile_rows(level), L.acmhip_packed_group_rows(level)
        if level in LEVELS:
            assert tr == 8192 >> level and gr in (16, 32) and tr % gr == 0 and gr % 4 == 0
            assert L.acmhip_packed_slots(level) % 4 == 0 and L.acmhip_packed_slots(level) * 64 >= (tr // 4) * (1 << level) // 2
        else:
            assert tr == 0 and gr == 0 and L.acmhip_packed_slots(level) == 0
            assert L.acmhip_pack_bound(level, 1, None) != 0


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("rows,pwr_max", [(1, 12), (3, 6), (16, 12), (17, 15), (700, 9)])
def test_pack_unpack_round_trip(level, rows, pwr_max):
    s, tr = staged_tiles(level, rows, 5, 31000 + level * 100 + rows, pwr_min=min(4, pwr_max), pwr_max=pwr_max,
                         val_max=65535 if pwr_max == 15 else 255)
    cols = 1 << level
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows * cols, level=level, rows=rows,
                        nrows=s.info.blocks * rows, row_begin=0)
    pk = capi.pack_streams(s.idx, [d])
    nt, slots = pk.streams[0].ntiles, capi.lib().acmhip_packed_slots(level)
    assert nt >= 5 and pk.chunks.size == nt * slots
    for k in range(nt):
        got = capi.unpack_tile(level, pk.chunks, pk.blob, k * slots)
        want = s.idx[k * tr * cols:(k + 1) * tr * cols].reshape(tr, cols)
        assert np.array_equal(got, want), (level, rows, k)
    # the form is smaller than the int16 arena it replaces, and every chunk is of one kind with a sane count
    assert pk.blob.nbytes + pk.chunks.nbytes < nt * tr * cols * 2 + 4096
    ch = pk.chunks[pk.chunks["kind"] != 0]
    assert set(np.unique(ch["kind"])) <= {1, 2, 3, 4} and ch["count"].min() >= 1
    gr = capi.lib().acmhip_packed_group_rows(level)
    assert ch["count"].max() <= 64 // (gr // 4) and (ch["row0"] % gr == 0).all()
    # the packer deals a tile's chunks out to the kernel's four waves evenly
    per_wave = (pk.chunks["kind"].reshape(nt, 4, slots // 4) != 0).sum(axis=2)
    assert (per_wave.max(axis=1) - per_wave.min(axis=1) <= 1).all()


@pytest.mark.parametrize("level", LEVELS)
def test_extreme_indices_choose_the_class(level):
    """hand-made index planes: all zero, the edges of every class in single column pairs, a full-range plane"""
    tr, cols = capi.packed_tile_rows(level), 1 << level
    gr = capi.lib().acmhip_packed_group_rows(level)
    idx = np.zeros((3 * tr, cols), dtype=np.int16)
    edges = [0, 1, -1, 7, -8, 8, -9, 127, -128, 128, -129, 32767, -32768]
    for k, v in enumerate(edges):
        idx[(k * 3) % tr, (2 * k) % cols + (k & 1)] = v          # tile 0: one value per column pair
    rng = np.random.default_rng(level)
    idx[tr:2 * tr] = rng.integers(-32768, 32768, size=(tr, cols), dtype=np.int64).astype(np.int16)
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=3 * tr * cols, level=level, rows=tr, nrows=3 * tr, row_begin=0)
    pk = capi.pack_streams(idx.reshape(-1), [d])
    assert pk.streams[0].ntiles == 3
    slots = capi.lib().acmhip_packed_slots(level)

    def chunks_of(k):
        c = pk.chunks[k * slots:(k + 1) * slots]
        return c[c["kind"] != 0]
    for k in range(3):
        assert np.array_equal(capi.unpack_tile(level, pk.chunks, pk.blob, k * slots), idx[k * tr:(k + 1) * tr])
    # tile 2 is all zeros: chunks of kind 1 only, no unit bytes beyond the column-pair lists
    ch = chunks_of(2)
    assert (ch["kind"] == 1).all() and ch["count"].sum() == (cols // 2) * (tr // gr)
    # tile 1 is full range: words everywhere
    assert (chunks_of(1)["kind"] == 4).all()
    # tile 0: the classes of the edge values (pair of column 2k holds edges[k] in its group); a chunk names a column pair by
    # where its first column sits in the kernel's padded LDS row: 2 p + p / 16
    kinds = {}
    for c in chunks_of(0):
        perm = pk.blob[int(c["blob_off16"]) * 16:][:2 * int(c["count"])].view("<u2")
        for at in perm:
            p = [q for q in range(cols // 2) if 2 * q + q // 16 == int(at)]
            assert len(p) == 1
            kinds[(int(c["row0"]), p[0])] = int(c["kind"])
    for k, v in enumerate(edges):
        row, pair = (k * 3) % tr, ((2 * k) % cols) // 2
        want = 1 if v == 0 else 2 if -8 <= v <= 7 else 3 if -128 <= v <= 127 else 4
        assert kinds[(row // gr * gr, pair)] == want, (v, kinds[(row // gr * gr, pair)])


def test_packer_rejects_what_it_cannot_take():
    L = capi.lib()
    nb = C.c_uint64()
    buf = np.zeros(1 << 16, dtype=np.uint8)
    assert L.acmhip_pack_tiles(5, buf.ctypes.data, 1, buf.ctypes.data, buf.ctypes.data, 0, C.byref(nb)) != 0
    assert L.acmhip_pack_tiles(9, buf.ctypes.data, 1, buf.ctypes.data, buf.ctypes.data, 8, C.byref(nb)) != 0
    # a corrupt table does not unpack: a chunk that names a column pair twice
    tr, cols = capi.packed_tile_rows(9), 512
    idx = np.ones((tr, cols), dtype=np.int16)
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=tr * cols, level=9, rows=tr, nrows=tr, row_begin=0)
    pk = capi.pack_streams(idx.reshape(-1), [d])
    assert np.array_equal(capi.unpack_tile(9, pk.chunks, pk.blob, 0), idx)
    bad = pk.blob.copy()
    off = int(pk.chunks[0]["blob_off16"]) * 16
    bad[off + 2:off + 4] = bad[off:off + 2]
    with pytest.raises(capi.AcmHipError):
        capi.unpack_tile(9, pk.chunks, bad, 0)
