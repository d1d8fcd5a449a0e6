"""GPU parity of the packed staged form: acm_tile2p (libacm_amd/csrc/acm_kernels.hip) fed by the host packer
(acm_pack.cpp) against the CPU oracle, bit-exact.  Reference semantics: decode.c:174-177 (set_pos: what is packed),
:181-502 (fillers: what ranges the packed indices have), :586-600 (value = idx * val), :508-577 (juggle_block).
"""
import numpy as np
import pytest

from helpers import fmt_args, make_stream, oracle_pcm
from libacm_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_k2(monkeypatch):
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | capi.PLAN_LEAN_ALWAYS)


def check_packed(dev, files, fmt=capi.FMT_S16LE, force_chans=0):
    staged = [capi.stage_file(f, force_chans) for f in files]
    got, st = capi.synth(dev, staged, fmt=fmt, return_stats=True, packed=True)
    be, sg = fmt_args(fmt)
    for k, (f, g) in enumerate(zip(files, got)):
        want, _ = oracle_pcm(f, force_chans, be, sg)
        assert g.size == want.size, (k, g.size, want.size)
        bad = np.nonzero(g != want)[0]
        assert bad.size == 0, "stream %d: %d/%d samples differ, first at %d (level %d rows %d)" % (
            k, bad.size, want.size, bad[0], staged[k].info.level, staged[k].info.rows)
    return st


@pytest.mark.parametrize("level", [6, 7, 8, 9])
@pytest.mark.parametrize("rows,pwr_max", [(1, 12), (3, 6), (16, 12), (16, 4), (17, 15), (700, 9)])
def test_packed_matrix(dev, force_k2, level, rows, pwr_max):
    """whole tiles from the packed form, the ragged tail from the int16 arena; block heights that make row values and
    width classes change inside a tile, inside a group and inside a row quad"""
    tr = 8192 >> level
    nblocks = max(2, (7 * tr + rows - 1) // rows + 1)
    f = make_stream(12000 + level * 100 + rows + pwr_max, level, rows, nblocks, cut=5, pwr_min=min(4, pwr_max), pwr_max=pwr_max,
                    val_max=65535 if pwr_max == 15 else 255)
    st = check_packed(dev, [f])
    assert st.packed_tiles >= 7 and st.fused_streams == 1 and st.stagewise_streams == 0


@pytest.mark.parametrize("fmt", [capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE])
def test_packed_batch(dev, force_k2, fmt):
    """many streams in one plan, levels with and without a packed form side by side: workgroup runs start inside streams
    (lead-in tiles) and cross stream boundaries; stereo; exact multiples of a tile and short tails"""
    files = []
    for i in range(41):
        lv = 5 + i % 8
        rows = [16, 5, 33, 1][i % 4]
        pm = [5, 12, 7, 15][(i // 2) % 4]
        files.append(make_stream(13000 + i, lv, rows, 2 + (i * 5) % 11 + ((16384 >> lv) * (1 + i % 3)) // rows,
                                 channels=1 + i % 2, cut=i % 3, pwr_min=min(4, pm), pwr_max=pm, val_max=65535 if i % 5 == 0 else 255))
    st = check_packed(dev, files, fmt=fmt)
    assert 0 < st.packed_tiles < st.tiles


def test_packed_every_filler_code(dev, force_k2):
    """every valid filler code in its own stream (26 of them: 0, 3-16 linear, the k / t codes), levels 6-9"""
    valid = [0] + list(range(3, 17)) + [17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29]
    files = []
    for j, code in enumerate(valid):
        lv = 6 + j % 4
        files.append(make_stream(14000 + j, lv, 16, 3 * (8192 >> lv) // 16 + 2, mix=2, single_code=code, pwr_min=15 if 3 <= code <= 16 else 4,
                                 pwr_max=15 if 3 <= code <= 16 else 12))
    check_packed(dev, files)


def test_packed_streams_with_h1_patches_keep_the_int16_form(dev, force_k2):
    """a stream with hazard-H1 patches never takes the packed build (its clean tiles stay with the general kernel); clean
    streams beside it do"""
    files = [make_stream(15000, 9, 16, 12, pwr_max=12),
             make_stream(15001, 9, 16, 12, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6),
             make_stream(15002, 7, 16, 40)]
    staged = [capi.stage_file(f) for f in files]
    assert staged[1].patches is not None and len(staged[1].patches) > 0
    st = check_packed(dev, files)
    assert st.packed_tiles == (12 * 16 // 16) + (40 * 16 // 64)


def test_packed_and_int16_launches_of_one_plan_agree(dev, force_k2):
    """acmhip_plan_bind_packed(NULL) sends the same plan back to the int16 arena: same PCM either way"""
    files = [make_stream(16000 + i, 6 + i % 4, 16, 30 + 7 * i, cut=i) for i in range(8)]
    staged = [capi.stage_file(f) for f in files]
    ar = capi.Arena(staged)
    pk = capi.pack_streams(ar.idx, ar.descs)
    d_idx, d_hdr, d_pcm = dev.malloc(ar.idx.nbytes), dev.malloc(ar.hdr.nbytes), dev.malloc(ar.pcm_words * 2)
    ptrs = pk.upload(dev)
    dev.upload(d_idx, ar.idx)
    dev.upload(d_hdr, ar.hdr)
    plan = capi.Plan(dev, ar.descs, packed=pk.streams)
    outs = []
    for bind in (ptrs, (None, None), ptrs):
        plan.bind_packed(*bind)
        dev.upload(d_pcm, np.zeros(ar.pcm_words, dtype=np.uint16))
        plan.launch(d_idx, d_hdr, d_pcm)
        o = np.zeros(ar.pcm_words, dtype=np.uint16)
        dev.download(o, d_pcm)
        outs.append(o)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    for f, (_, _, po, _, ne) in zip(files, ar.layout):
        assert np.array_equal(outs[0][po:po + ne], oracle_pcm(f)[0])
    with pytest.raises(capi.AcmHipError):
        plan.bind_packed(ptrs[0], None)
    plan.destroy()
    for p in (d_idx, d_hdr, d_pcm) + ptrs:
        dev.free(p)


@pytest.mark.parametrize("prestage", [False, True])
def test_batch_decode_stages_packed(dev, prestage):
    """acm_batch_decode with ACM_BATCH_STAGE_PACKED: the host pool packs the whole tiles of every clean stream of a level that
    has the form, the upload carries the packed form and the int16 rows of the ragged tails only - same PCM and statuses as the
    oracle for clean, ragged, stereo, truncated, H1-patched, tiny and non-ACM files of every level, fewer bytes over PCIe"""
    import oracle_api as O
    files = [make_stream(17000 + i, [7, 9, 5, 8, 11, 6, 13, 9][i % 8], [16, 3, 1, 33][i % 4], 3 + (i * 7) % 23 + (8192 >> [7, 9, 5, 8, 11, 6, 13, 9][i % 8]) // 4,
                         channels=1 + i % 2, cut=i % 5, pwr_max=[12, 6, 15][i % 3], val_max=65535 if i % 3 == 2 else 255) for i in range(61)]
    files[5] = files[5][:len(files[5]) * 2 // 3]
    files[11] = b"RIFFnope"
    files[17] = make_stream(17990, 7, 16, 40, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    files[23] = make_stream(17991, 9, 16, 1)                 # one block = one tile exactly: nothing travels as int16
    files[29] = make_stream(17992, 9, 16, 64, mix=2, single_code=16, pwr_min=15, pwr_max=15)     # 16-bit indices throughout
    plain, tm0 = capi.batch_decode(dev, files, threads=4, prestage=prestage, byteplane=False)
    res, tm = capi.batch_decode(dev, files, threads=4, prestage=prestage, packed=True)
    assert tm0.packed_streams == 0 and tm.packed_streams >= 25, (tm0.packed_streams, tm.packed_streams)
    # (pieces of one arena closer together than a transfer call is worth travel as one, the bytes between them included: with streams this
    # small a good part of the int16 rows the packed form replaces travels anyway)
    assert tm.h2d_bytes < 0.9 * tm0.h2d_bytes, (tm.h2d_bytes, tm0.h2d_bytes)
    for k, f in enumerate(files):
        o = O.Oracle(f)
        if o.err < 0:
            assert res[k][0] == o.err and res[k][1].size == 0, k
            continue
        want, wst = oracle_pcm(f)
        assert res[k][0] == wst and np.array_equal(res[k][1], want), k
        assert plain[k][0] == wst and np.array_equal(plain[k][1], want), k
    # the device parser stages int16: the flag is ignored there
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, packed=True)
    assert tm.packed_streams == 0
    for k, f in enumerate(files):
        if O.Oracle(f).err >= 0:
            assert np.array_equal(res[k][1], oracle_pcm(f)[0]), k
