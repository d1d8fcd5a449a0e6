"""GPU parity of the byte-plane staged form: acm_tile2's matrix-core build (libacm_amd/csrc/acm_kernels.hip, FirstPassM: the first
three stages of juggle_block, reference decode.c:527-590, as one v_mfma_i32_16x16x32_i8 per row pair and 16 residue classes) fed by
the host stager (acmhip_mform_rows, acm_pack.cpp) against the CPU oracle, bit-exact.
"""
import numpy as np
import pytest

from helpers import fmt_args, make_stream, oracle_pcm
from libacm_amd import capi

pytestmark = pytest.mark.gpu

LEVELS = [7, 8, 9, 10, 11, 12, 13, 14]


@pytest.fixture
def force_k2(monkeypatch):
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | capi.PLAN_LEAN_ALWAYS)


def tile_rows(level):
    return capi.lib().acmhip_mform_tile_rows(level)


def plan_rows(level):
    """rows the planner hands out at a time: whole tiles of the lean kernel's vector-ALU build, cut into the byte-plane build's own (a level
    of the chunk kernel: 2048-sample chunks; level 13: row pairs)"""
    return max(tile_rows(level), capi.lib().acmk_tile2_rows(level), 4)


def check(dev, files, fmt=capi.FMT_S16LE, force_chans=0):
    staged = [capi.stage_file(f, force_chans) for f in files]
    got, st = capi.synth(dev, staged, fmt=fmt, return_stats=True, mform=True)
    be, sg = fmt_args(fmt)
    for k, (f, g) in enumerate(zip(files, got)):
        want, _ = oracle_pcm(f, force_chans, be, sg)
        assert g.size == want.size, (k, g.size, want.size)
        bad = np.nonzero(g != want)[0]
        assert bad.size == 0, "stream %d: %d/%d samples differ, first at %d (level %d rows %d)" % (
            k, bad.size, want.size, bad[0], staged[k].info.level, staged[k].info.rows)
    return st


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("rows,pwr_max", [(1, 12), (2, 9), (3, 6), (16, 12), (16, 4), (17, 15), (700, 9)])
def test_byteplane_matrix(dev, force_k2, level, rows, pwr_max):
    """whole tiles from the byte-plane form, the ragged tail from the int16 arena; block heights that put a val change between the
    rows of a unit in every possible place (even: between row pairs, odd: inside them, 1: everywhere) and 16-bit indices with
    16-bit row values (pwr_max 15)"""
    tr = plan_rows(level)
    nblocks = max(2, (7 * tr + rows - 1) // rows + 1)
    f = make_stream(22000 + level * 100 + rows + pwr_max, level, rows, nblocks, cut=5, pwr_min=min(4, pwr_max), pwr_max=pwr_max,
                    val_max=65535 if pwr_max == 15 else 255)
    st = check(dev, [f])
    if level >= 13 and int(capi.stage_file(f).idx.max()) >= 32640:
        assert st.mform_tiles == 0            # two signed bytes end at 32639: at levels 13 / 14 such a stream stays int16 (levels 8-12 have a whole-range class)
    else:
        assert st.mform_tiles >= 7
    assert st.fused_streams == 1 and st.stagewise_streams == 0


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("rows,pwr_min,pwr_max", [(16, 3, 3), (16, 3, 12), (3, 3, 7), (1, 3, 12), (6, 3, 5), (64, 3, 9)])
def test_byteplane_width_classes(dev, force_k2, level, rows, pwr_min, pwr_max):
    """quiet blocks travel at 4 or 8 bits per index: row pairs of every width, width changes between the rows of a unit in every place
    a block boundary can fall (with and without a change of val), a stream that is 4 bits throughout.  (pwr >= 3: below that the k / t
    fillers' indices of up to 5 leave the amplitude table - H1 patches - and such a stream keeps the int16 form)"""
    tr = plan_rows(level)
    nblocks = max(2, (9 * tr + rows - 1) // rows + 1)
    f = make_stream(29000 + level * 100 + rows + pwr_max, level, rows, nblocks, cut=2, pwr_min=pwr_min, pwr_max=pwr_max)
    s = capi.stage_file(f)
    assert s.patches is None or len(s.patches) == 0
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows << level, level=level, rows=rows, nrows=s.info.blocks * rows, row_begin=0)
    cc = capi.mform_streams(s.idx, [d]).class_counts()
    narrow = 2 if capi.lib().acmhip_mform_group(level) == 64 else 1           # (the chunk kernel's form has no 4-bit class)
    if pwr_max <= 3:
        assert cc[3] == 0 and cc[narrow] > 9 and cc[1] + cc[2] == cc[narrow]
    elif pwr_max == 12 and rows < 64:
        assert int(cc[1] > 1) + int(cc[2] > 1) + int(cc[3] > 0) >= 2, cc          # at least two widths among its blocks
    st = check(dev, [f])
    assert st.mform_tiles >= 9


@pytest.mark.parametrize("level", [8, 9, 10, 11, 12])
@pytest.mark.parametrize("rows,pwr_min,pwr_max,val_max", [(16, 8, 10, 255), (2, 8, 10, 255), (64, 9, 9, 65535), (16, 6, 12, 255), (1, 6, 12, 255),
                                                           (6, 7, 11, 65535), (4, 8, 10, 65535), (250, 10, 10, 255)])
def test_byteplane_twelve_bit_class(dev, force_k2, level, rows, pwr_min, pwr_max, val_max):
    """levels 8-12: pairs whose indices fit a signed low byte and a signed high nibble travel at 12 bits (pwr 8-10).  Streams that are 12
    bits throughout (the chunk kernel's fast path on nibble planes: one val per chunk in the middle of a block, the general path with
    its conversions at every block boundary), every neighbourhood of 8 / 12 / 16-bit pairs a block boundary can produce (pwr 6-12, block
    heights 1 ... 16), and row values beyond 16 bits as scaled (val_max 65535: the general path at levels 10-12) - against the CPU oracle"""
    tr = plan_rows(level)
    nblocks = max(3, (11 * tr + rows - 1) // rows + 1)
    f = make_stream(31000 + level * 100 + rows + pwr_max, level, rows, nblocks, cut=3, pwr_min=pwr_min, pwr_max=pwr_max, val_max=val_max)
    s = capi.stage_file(f)
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows << level, level=level, rows=rows, nrows=s.info.blocks * rows, row_begin=0)
    cc = capi.mform_streams(s.idx, [d]).class_counts()
    if pwr_min >= 8 and pwr_max <= 10:
        assert cc[1] > 0 and cc[3] == 0, cc             # (a pair may happen to fit 8 bits)
    elif pwr_min == 6 and pwr_max == 12 and rows <= 16:
        assert cc[1] > 0, cc
    st = check(dev, [f])
    assert st.mform_tiles >= 9


@pytest.mark.parametrize("level", [8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("rows,val_max", [(16, 255), (3, 65535), (1, 65535), (64, 255)])
def test_byteplane_whole_range_class(dev, force_k2, level, rows, val_max):
    """VERDICT r5, task 3 (iii): an index beyond 32639 (two signed bytes end there; it takes pwr 15) no longer keeps a stream
    out of the form: such PAIRS are written with the unsigned low byte (class code 0, stored minus 128) and the kernels - the chunk
    kernel's general path at levels 8-12, FirstPassZW inside acm_tile2 at 13 / 14 (row sums from the matrix cores themselves) - add
    128 x val x the coefficient row sums back, row by row.  Blocks of pwr 15 beside quiet ones (every neighbourhood of the
    whole-range class with 8 / 12 / 16-bit pairs), block boundaries inside a chunk, row values of 8 and of 16 bits"""
    tr = plan_rows(level)
    nblocks = max(4, (9 * tr + rows - 1) // rows + 1)
    f = make_stream(34000 + level * 100 + rows, level, rows, nblocks, cut=1, pwr_min=5, pwr_max=15, val_max=val_max)
    s = capi.stage_file(f)
    if int(s.idx.max()) < 32640:
        f = make_stream(34500 + level * 100 + rows, level, rows, nblocks, cut=1, mix=2, single_code=16, pwr_min=15, pwr_max=15, val_max=val_max)
        s = capi.stage_file(f)
    assert int(s.idx.max()) >= 32640
    d = capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=0, n_emit=s.info.blocks * rows << level, level=level, rows=rows, nrows=s.info.blocks * rows, row_begin=0)
    mf = capi.mform_streams(s.idx, [d])
    npairs = mf.streams[0].ntiles * tile_rows(level) // 2
    assert npairs > 0 and ((mf.pairs[1:1 + npairs] & 3) == 0).any()
    st = check(dev, [f])
    assert st.mform_tiles >= 9


@pytest.mark.parametrize("fmt", [capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE])
def test_byteplane_batch(dev, force_k2, fmt):
    """many streams in one plan, levels with and without the form side by side: workgroup runs start inside streams (lead-in tiles)
    and cross stream boundaries; stereo; exact multiples of a tile and short tails"""
    files = []
    for i in range(45):
        lv = 5 + i % 9
        rows = [16, 5, 33, 1][i % 4]
        pm = [5, 12, 7, 15][(i // 2) % 4]
        files.append(make_stream(23000 + i, lv, rows, 2 + (i * 5) % 11 + ((32768 >> lv) * (1 + i % 3)) // rows,
                                 channels=1 + i % 2, cut=i % 3, pwr_min=min(4, pm), pwr_max=pm, val_max=65535 if i % 5 == 0 else 255))
    st = check(dev, files, fmt=fmt)
    assert 0 < st.mform_tiles < st.tiles


def test_byteplane_every_filler_code(dev, force_k2):
    """every valid filler code in its own stream (26 of them: 0, 3-16 linear, the k / t codes), levels 7-12"""
    valid = [0] + list(range(3, 17)) + [17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29]
    files = []
    for j, code in enumerate(valid):
        lv = 7 + j % 8
        files.append(make_stream(24000 + j, lv, 16, 3 * plan_rows(lv) // 16 + 2, mix=2, single_code=code, pwr_min=15 if 3 <= code <= 16 else 4,
                                 pwr_max=15 if 3 <= code <= 16 else 12))
    check(dev, files)


def test_byteplane_extreme_indices(dev, force_k2):
    """the widest linear filler with the largest row values: indices over the whole int16 range (the high byte plane at -128 and 127,
    the low one at both ends), val = 65535"""
    files = [make_stream(24500 + lv, lv, 16, 4 * plan_rows(lv) // 16 + 1, mix=2, single_code=16, pwr_min=15, pwr_max=15, val_max=65535) for lv in LEVELS]
    staged = [capi.stage_file(f) for f in files]
    assert max(int(s.idx.max()) for s in staged) > 32000 and min(int(s.idx.min()) for s in staged) < -32000
    check(dev, files)


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("rows", [16, 2])
def test_byteplane_windows(dev, force_k2, level, rows):
    """windows into a stream that came with a byte-plane form (what a block range of a device-parsed batch is): a window that starts on a
    tile boundary goes to the lean kernels behind a lead-in record for the tile in front of it (ACM_TILE_DISCARD: decoded for its carries,
    stored into the sink), its ragged tail and every other window to the general tile kernel - same PCM as the oracle's from that row on,
    with the form bound and, through the int16 twins of the same records, without"""
    t2 = capi.lib().acmk_tile2_rows(level)
    nblocks = (9 * t2 + 5 + rows - 1) // rows
    f = make_stream(25500 + level * 10 + rows, level, rows, nblocks, cut=3, pwr_max=11)
    st = capi.stage_file(f)
    if level > 12:
        pytest.skip("levels 13 / 14 leave the lean kernels to whole-stream plans")
    cols = 1 << level
    ar = capi.Arena([st])
    mf = capi.mform_streams(ar.idx, ar.descs)
    assert mf.streams[0].ntiles > 0
    want, _ = oracle_pcm(f)
    nrows = st.info.blocks * rows
    begins = [0, t2, 2 * t2, 5 * t2, t2 + 1, 8 * t2]
    descs, at = [], 0
    for rb in begins:
        ne = want.size - rb * cols
        descs.append(capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=at, n_emit=ne, level=level, rows=rows, nrows=nrows, row_begin=rb))
        at += (ne + 63) // 64 * 64
    d_idx, d_hdr, d_pcm = dev.malloc(ar.idx.nbytes), dev.malloc(ar.hdr.nbytes), dev.malloc(at * 2)
    d_mf = mf.upload(dev)
    dev.upload(d_idx, ar.idx)
    dev.upload(d_hdr, ar.hdr)
    plan = capi.Plan(dev, descs, packed=[mf.streams[0]] * len(descs))
    try:
        for k, rb in enumerate(begins):
            whole = (min(nrows - rb, (want.size - rb * cols) >> level) // t2) * t2
            assert plan.form_rows(k) == (whole if rb % t2 == 0 else 0), (k, rb)
        for bind in (d_mf, (None, None)):
            plan.bind_mform(*bind)
            dev.upload(d_pcm, np.full(at, 0xA5A5, dtype=np.uint16))
            plan.launch(d_idx, d_hdr, d_pcm)
            got = np.zeros(at, dtype=np.uint16)
            dev.download(got, d_pcm)
            for k, rb in enumerate(begins):
                d = descs[k]
                assert np.array_equal(got[d.pcm_off:d.pcm_off + d.n_emit].view(np.int16), want[rb * cols:].view(np.int16)), (bind[0] is not None, k, rb)
    finally:
        plan.destroy()
        for p_ in (d_idx, d_hdr, d_pcm) + d_mf:
            dev.free(p_)


@pytest.mark.parametrize("level", [9, 10, 11, 12, 13, 14])
def test_byteplane_val_around_the_one_instruction_join(dev, force_k2, level):
    """the high plane of a 16-bit pair joins in one v_mad_u32_u24 per output while the row's scaled val (val << (16 - level) from level 10 on)
    stays below 2^16, in three instructions above it and wherever val changes in reach: blocks on both sides of that border, next to each other"""
    border = 65536 >> (16 - level if level >= 10 else 0)
    rows = 16 if level < 13 else 4
    for k, vmax in enumerate((border - 1, 2 * border - 1 if 2 * border <= 65536 else 65535, 65535)):
        f = make_stream(29500 + 10 * level + k, level, rows, max(6, (6 * 8192 >> level) // rows + 2), pwr_min=8, pwr_max=12, val_max=vmax)
        got, st = capi.synth(dev, [capi.stage_file(f)], return_stats=True, mform=True)
        assert st.mform_tiles > 0
        assert np.array_equal(got[0], oracle_pcm(f)[0]), (level, vmax)


def test_byteplane_streams_with_h1_patches_keep_the_int16_form(dev, force_k2):
    files = [make_stream(25000, 9, 16, 12, pwr_max=12),
             make_stream(25001, 9, 16, 12, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6),
             make_stream(25002, 11, 16, 40)]
    staged = [capi.stage_file(f) for f in files]
    assert staged[1].patches is not None and len(staged[1].patches) > 0
    st = check(dev, files)
    assert st.mform_tiles == (12 * 16 // tile_rows(9)) + (40 * 16 // tile_rows(11))


def test_byteplane_and_int16_launches_of_one_plan_agree(dev, force_k2):
    """acmhip_plan_bind_mform(NULL) sends the same plan back to the int16 arena: same PCM either way"""
    files = [make_stream(26000 + i, 7 + i % 6, 16, 30 + 7 * i, cut=i) for i in range(9)]
    staged = [capi.stage_file(f) for f in files]
    ar = capi.Arena(staged)
    mf = capi.mform_streams(ar.idx, ar.descs)
    d_idx, d_hdr, d_pcm = dev.malloc(ar.idx.nbytes), dev.malloc(ar.hdr.nbytes), dev.malloc(ar.pcm_words * 2)
    d_mf = mf.upload(dev)
    dev.upload(d_idx, ar.idx)
    dev.upload(d_hdr, ar.hdr)
    plan = capi.Plan(dev, ar.descs, packed=mf.streams)
    assert plan.stats().mform_tiles > 0
    outs = []
    for bind in (d_mf, (None, None), d_mf):
        plan.bind_mform(*bind)
        dev.upload(d_pcm, np.zeros(ar.pcm_words, dtype=np.uint16))
        plan.launch(d_idx, d_hdr, d_pcm)
        o = np.zeros(ar.pcm_words, dtype=np.uint16)
        dev.download(o, d_pcm)
        outs.append(o)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    for f, (_, _, po, _, ne) in zip(files, ar.layout):
        assert np.array_equal(outs[0][po:po + ne], oracle_pcm(f)[0])
    with pytest.raises(capi.AcmHipError):
        plan.bind_mform(d_mf[0], None)
    plan.destroy()
    for p in (d_idx, d_hdr, d_pcm) + d_mf:
        dev.free(p)


def test_plan_modifiers_form_only_and_async_upload(dev, force_k2):
    """ACMHIP_PLAN_FORM_ONLY: the int16 twin of the byte-plane records is not cut - a launch without the form bound fails loudly instead
    of reading rows that may never have been staged; ACMHIP_PLAN_UPLOAD_ASYNC: the tables are only queued when the plan comes back,
    its launches wait for them on the device.  Same PCM as the oracle either way (what acm_batch_decode's plans are made with)"""
    files = [make_stream(26300 + i, 8 + i % 5, 16, 64 + 11 * i, cut=i, pwr_max=12) for i in range(8)]
    staged = [capi.stage_file(f) for f in files]
    ar = capi.Arena(staged)
    mf = capi.mform_streams(ar.idx, ar.descs)
    # only the rows the plan does not read from the form exist as int16 on the device: everything else is poison
    probe = capi.Plan(dev, ar.descs, packed=mf.streams)
    idx = np.full(ar.idx.size, 0x5A5A, dtype=np.int16)
    for i, d in enumerate(ar.descs):
        r2 = probe.form_rows(i)
        keep = max(0, r2 - 2) << d.level
        idx[d.idx_off + keep:d.idx_off + (d.nrows << d.level)] = ar.idx[d.idx_off + keep:d.idx_off + (d.nrows << d.level)]
    probe.destroy()
    d_idx, d_hdr, d_pcm = dev.malloc(ar.idx.nbytes), dev.malloc(ar.hdr.nbytes), dev.malloc(ar.pcm_words * 2)
    d_mf = mf.upload(dev)
    dev.upload(d_idx, idx)
    dev.upload(d_hdr, ar.hdr)
    for flags in (capi.PLAN_FORM_ONLY, capi.PLAN_UPLOAD_ASYNC, capi.PLAN_FORM_ONLY | capi.PLAN_UPLOAD_ASYNC):
        plan = capi.Plan(dev, ar.descs, flags=flags, packed=mf.streams)
        assert plan.stats().mform_tiles > 0
        if flags & capi.PLAN_FORM_ONLY:
            with pytest.raises(capi.AcmHipError):
                plan.launch(d_idx, d_hdr, d_pcm)            # no form bound, no twin to fall back to
        plan.bind_mform(*d_mf)
        dev.upload(d_pcm, np.full(ar.pcm_words, 0xA5A5, dtype=np.uint16))
        plan.launch(d_idx, d_hdr, d_pcm)
        out = np.zeros(ar.pcm_words, dtype=np.uint16)
        dev.download(out, d_pcm)
        for f, (_, _, po, _, ne) in zip(files, ar.layout):
            assert np.array_equal(out[po:po + ne], oracle_pcm(f)[0]), flags
        plan.destroy()
    for p in (d_idx, d_hdr, d_pcm) + d_mf:
        dev.free(p)


def test_plan_rejects_a_byteplane_form_that_is_too_short(dev, force_k2):
    """a stream that comes with fewer byte-plane tiles than the plan would decode from them, or with an unknown form, is an argument
    error at plan creation - not a launch that reads past its pair table"""
    f = make_stream(26500, 9, 16, 6)
    s = capi.stage_file(f)
    ar = capi.Arena([s])
    mf = capi.mform_streams(ar.idx, ar.descs)
    nt = 6 * 16 // tile_rows(9)
    assert mf.streams[0].ntiles == nt
    for ntiles, form in ((nt - 1, capi.FORM_BYTEPLANE), (nt, 7)):
        with pytest.raises(capi.AcmHipError):
            capi.Plan(dev, ar.descs, packed=[capi.PackedStream(0, ntiles, form)])
    capi.Plan(dev, ar.descs, packed=[capi.PackedStream(0, nt, capi.FORM_BYTEPLANE)]).destroy()


@pytest.mark.parametrize("prestage", [False, True])
def test_batch_decode_stages_byteplanes(dev, prestage):
    """acm_batch_decode with ACM_BATCH_STAGE_BYTEPLANE: the host pool re-orders the whole tiles of every clean stream of a level
    that has the form, the upload carries that and the int16 rows of the ragged tails only - same PCM and statuses as the oracle
    for clean, ragged, stereo, truncated, H1-patched, tiny and non-ACM files of every level"""
    import oracle_api as O
    lv = [7, 9, 5, 8, 11, 6, 13, 10, 12]
    files = [make_stream(27000 + i, lv[i % 9], [16, 3, 1, 33][i % 4], 3 + (i * 7) % 23 + (16384 >> lv[i % 9]) // 4,
                         channels=1 + i % 2, cut=i % 5, pwr_max=[12, 6, 15][i % 3], val_max=65535 if i % 3 == 2 else 255) for i in range(63)]
    files[5] = files[5][:len(files[5]) * 2 // 3]
    files[11] = b"RIFFnope"
    files[17] = make_stream(27990, 7, 16, 40, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    files[23] = make_stream(27991, 9, 16, 1)                 # one block = one tile exactly: nothing travels as int16
    plain, tm0 = capi.batch_decode(dev, files, threads=4, prestage=prestage, byteplane=False)
    res, tm = capi.batch_decode(dev, files, threads=4, prestage=prestage, byteplane=True)
    dflt, tmd = capi.batch_decode(dev, files, threads=4, prestage=prestage)         # the default: the form, unless the blocks were parsed ahead
    assert tmd.packed_streams == (0 if prestage else tm.packed_streams)
    assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(dflt, res))
    assert tm0.packed_streams == 0 and tm.packed_streams >= 30, (tm0.packed_streams, tm.packed_streams)
    assert tm.h2d_bytes < 0.95 * tm0.h2d_bytes, (tm.h2d_bytes, tm0.h2d_bytes)          # quiet blocks travel at 4 or 8 bits per index
    for k, f in enumerate(files):
        o = O.Oracle(f)
        if o.err < 0:
            assert res[k][0] == o.err and res[k][1].size == 0, k
            continue
        want, wst = oracle_pcm(f)
        # the batch status is what stopped the parser; an acm_read_loop() caller may see that error swallowed (test_gpu_parity.py)
        assert (res[k][0] == wst or (res[k][0] < 0 and wst <= 0)) and np.array_equal(res[k][1], want), k
        assert plain[k][0] == res[k][0] and np.array_equal(plain[k][1], want), k
    # the device parser writes the form itself for the streams that can have it there (the chunk kernel's levels, even block heights)
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, byteplane=True)
    assert 0 < tm.packed_streams < tm0.packed_streams + 63
    for k, f in enumerate(files):
        if O.Oracle(f).err >= 0:
            assert np.array_equal(res[k][1], oracle_pcm(f)[0]), k


@pytest.mark.parametrize("level", [8, 9, 10, 11, 12])
def test_batch_stages_odd_block_heights(dev, level):
    """VERDICT r5, task 3 (ii): streams whose acm_rows is odd - row pairs that straddle block boundaries, a val change INSIDE a pair - get
    the byte-plane form from the host parsing pass itself (acm_stage_file_mform) and are decoded from it by the chunk kernel: block
    heights 1, 3, 17, 33, mono and stereo, ragged ends, all three width classes"""
    files = [make_stream(33000 + 50 * level + i, level, rows, max(3, (9 * plan_rows(level) + rows - 1) // rows + i), channels=1 + i % 2, cut=i,
                         pwr_min=[4, 8, 6, 12][i % 4], pwr_max=[12, 10, 9, 12][i % 4])
             for i, rows in enumerate([1, 3, 17, 33, 3, 1, 33, 17])]
    res, tm = capi.batch_decode(dev, files, threads=4)
    assert tm.packed_streams == len(files), tm.packed_streams               # every one of them travelled in the form
    plain, tm0 = capi.batch_decode(dev, files, threads=4, byteplane=False)
    assert tm0.packed_streams == 0 and tm.h2d_bytes < tm0.h2d_bytes
    for k, f in enumerate(files):
        want, wst = oracle_pcm(f)
        assert res[k][0] == wst and np.array_equal(res[k][1], want), (level, k)
        assert np.array_equal(plain[k][1], want), (level, k)
    # and through the plan API, with the form staged in two passes (acm_stage_file + acmhip_mform_rows): tiles really come from the form
    st = check(dev, files[:4])
    assert st.mform_tiles >= 4 * 9


def test_batch_first_call_on_a_fresh_device_with_small_high_level_streams():
    """a byte-plane batch must not rely on int16 rows an EARLIER call left in the device arena.  A few level-13 / 14 streams are
    too few tiles for the lean kernel (the plan sends them to the prefix + plane pair, which reads the int16 arena from row 0): their
    int16 rows have to travel although a byte-plane block was staged for them.  First call on a device handle of its own, nothing decoded
    before it; then once more behind a decoy batch of other files of the same sizes."""
    import oracle_api as O
    files = [make_stream(27500 + i, [13, 14, 13, 9, 14, 11][i % 6], [4, 2, 6, 16, 3, 8][i % 6], 6 + i % 3, pwr_max=9) for i in range(12)]
    decoy = [make_stream(27600 + i, [13, 14, 13, 9, 14, 11][i % 6], [4, 2, 6, 16, 3, 8][i % 6], 6 + i % 3, pwr_max=9) for i in range(12)]
    fresh = capi.Device(0)
    try:
        for batch in (files, decoy, files):
            res, tm = capi.batch_decode(fresh, batch, threads=2, byteplane=True)
            assert tm.packed_streams >= 4
            for k, f in enumerate(batch):
                assert np.array_equal(res[k][1], oracle_pcm(f)[0]), k
    finally:
        fresh.close()


@pytest.mark.parametrize("ranges", [1, 3, 16])
def test_device_parser_stages_byteplanes(dev, monkeypatch, ranges):
    """VERDICT r4, task 3 (i): acm_batch_decode with device parsing AND ACM_BATCH_STAGE_BYTEPLANE - the column kernel writes the chunk kernel's
    form itself (width from the block's pwr, places from a running sum in the walk; only the rows behind the whole tiles exist as int16) and
    the batch's plans - one per chunk, or one per block range, whose streams are windows - read it on the lean kernels.  Every valid filler
    code at levels 8-12, block heights 2 / 16 / 64 / 700, quiet and loud blocks, stereo; truncated, H1, junk and odd-height files fall back.
    Same PCM and statuses as the oracle; and the same with the int16 staging of the device parser (ACM_BATCH_DEV_MFORM=0)"""
    import oracle_api as O
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    valid = [0] + list(range(3, 17)) + [17, 18, 19, 20, 21, 22, 23, 24, 26, 27, 29]
    files = []
    for j, code in enumerate(valid):
        lv = 8 + j % 5
        rows = [16, 2, 64, 16, 700][j % 5]
        nblocks = max(3, (6 * capi.lib().acmk_tile2_rows(lv) + rows - 1) // rows + 1 + j % 3)
        files.append(make_stream(30000 + j, lv, rows, nblocks, mix=2, single_code=code, channels=1 + j % 2, cut=j % 4,
                                 pwr_min=14 if 3 <= code <= 16 else 4, pwr_max=14 if 3 <= code <= 16 else 12))
    for i in range(30):
        lv = [9, 11, 10, 7, 12, 8, 13, 9, 5][i % 9]
        files.append(make_stream(30100 + i, lv, [16, 3, 64, 8][i % 4], 4 + (i * 7) % 23 + (16384 >> lv) // 4, channels=1 + i % 2, cut=i % 5,
                                 pwr_max=[12, 6, 15][i % 3], val_max=65535 if i % 3 == 2 else 255))
    files[3] = files[3][:len(files[3]) * 2 // 3]
    files[9] = b"RIFFnope"
    files[13] = make_stream(30990, 9, 16, 40, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    files[21] = make_stream(30991, 9, 16, 1)
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, byteplane=True)
    # (a range ends on a tile boundary of the lean kernel whatever the block height - acmk_range_bound -, so the form does not depend on the range count)
    assert tm.device_parsed >= 40 and tm.packed_streams >= 25, (tm.device_parsed, tm.packed_streams)
    res0, tm0 = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, byteplane=False)      # ACM_BATCH_STAGE_INT16: every row as int16
    assert tm0.packed_streams == 0
    for k, f in enumerate(files):
        o = O.Oracle(f)
        if o.err < 0:
            assert res[k][0] == o.err and res[k][1].size == 0, k
            continue
        want, wst = oracle_pcm(f)
        assert (res[k][0] == wst or (res[k][0] < 0 and wst <= 0)) and np.array_equal(res[k][1], want), k
        assert res0[k][0] == res[k][0] and np.array_equal(res0[k][1], want), k


@pytest.mark.parametrize("ranges", [1, 4])
def test_device_parser_stages_every_width_class_and_odd_block_heights(dev, monkeypatch, ranges):
    """The column kernel of the device parser writes three of the form's width classes - 8 bits, two signed bytes, and the whole-range
    class for pwr 15 blocks at levels 8-12 (no host redo for an index beyond 32 639 there) - and
    block heights that are ODD: every other block begins inside a row pair, which takes the wider class of the two blocks and whose
    second row is written by the next block's threads (acm_parse.hip bp_class / bp_wider).  Same PCM as the oracle, and the streams
    really travel in the form - in one piece and in block ranges"""
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    files = []
    for j, (lv, rows) in enumerate(((8, 1), (8, 3), (9, 5), (9, 17), (10, 3), (10, 33), (11, 1), (11, 7), (12, 3), (12, 9), (13, 3), (13, 5), (14, 1), (14, 3),
                                    (9, 16), (10, 8), (12, 4), (8, 64), (11, 2), (9, 2))):
        tr = capi.lib().acmk_tile2_rows(lv)
        nblocks = max(4, (5 * tr + rows - 1) // rows + 2 + j % 3)
        loud = j % 4 == 1 and lv <= 12                  # pwr 15 with 16-bit values: indices beyond 32 639 occur
        files.append(make_stream(36000 + j, lv, rows, nblocks, channels=1 + j % 2, cut=j % 5, pwr_min=[4, 13, 8, 6][j % 4],
                                 pwr_max=15 if loud else [12, 15, 10, 11][j % 4] if lv <= 12 else [12, 14, 10, 11][j % 4],
                                 val_max=65535 if j % 3 == 0 else 255))
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, byteplane=True)
    assert tm.device_parsed == len(files) and tm.host_parsed == 0, (tm.device_parsed, tm.host_parsed)
    for k, f in enumerate(files):
        want, wst = oracle_pcm(f)
        assert res[k][0] == wst and np.array_equal(res[k][1], want), k
    # (the level-13 / 14 streams of a batch this small are below what the lean kernels take: they are staged as int16, like on the host route)
    assert tm.packed_streams >= 16, tm.packed_streams


@pytest.mark.parametrize("ranges", [1, 3, 7])
@pytest.mark.parametrize("level", [8, 9, 10, 11, 12])
def test_device_parser_stages_odd_block_heights(dev, monkeypatch, level, ranges):
    """the same batches as test_batch_stages_odd_block_heights, parsed AND staged on the device: every stream travels in the form, same PCM
    as the oracle.  (Levels 13 / 14: both fused stagers - this one and acm_stage_file_mform - keep the int16 form, because a plan too small
    for the lean kernel reads int16 rows from row 0 on; acm_batch.cpp, acm_stream.cpp.)  In one piece and in block ranges: a stream's
    ranges are cut where whole tiles end (acmk_range_bound), so no range cuts a row pair or leaves a ragged end inside the form."""
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    files = [make_stream(33000 + 50 * level + i, level, rows, max(3, (9 * plan_rows(level) + rows - 1) // rows + i), channels=1 + i % 2, cut=i,
                         pwr_min=[4, 8, 6, 12][i % 4], pwr_max=[12, 10, 9, 12][i % 4])
             for i, rows in enumerate([1, 3, 17, 33, 3, 1, 33, 17])]
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE, byteplane=True)
    assert tm.device_parsed == len(files) and tm.host_parsed == 0 and tm.packed_streams == len(files), (tm.device_parsed, tm.host_parsed, tm.packed_streams)
    for k, f in enumerate(files):
        want, wst = oracle_pcm(f)
        assert res[k][0] == wst and np.array_equal(res[k][1], want), (level, k)


@pytest.mark.parametrize("ranges", [2, 3, 5])
def test_device_parser_walk_that_stops_inside_a_range(dev, monkeypatch, ranges):
    """A truncated stream whose walk gets through SOME blocks of a block range and stops in a later one: the column kernel then skips the
    stream altogether, so none of that range's pair-table entries is written by it - and the synthesis of the range is already queued
    on a plan cut from the headers.  Every entry of the range must name a place the chunk kernel may load from (it used to be only the
    entries from the block that failed on: a GPU memory fault, profiles/byteplane_fuzz.py seed 2718 batch 133).  The host reader's
    redo delivers what the reference delivers"""
    import oracle_api as O
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    whole = make_stream(500842608, 12, 8, 3, channels=1, cut=5, pwr_min=7, pwr_max=7, val_max=65535)
    files = [whole[:36185]]
    for k, (lv, rows, nb) in enumerate(((12, 8, 7), (9, 16, 12), (10, 8, 9), (11, 4, 11), (8, 32, 6))):
        f = make_stream(33100 + k, lv, rows, nb, pwr_max=12)
        for frac in (0.55, 0.72, 0.9):
            files.append(f[:int(len(f) * frac)])
    files.append(make_stream(33200, 9, 16, 12, pwr_max=12))        # a clean one beside them
    for _ in range(2):
        res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE)
        for k, f in enumerate(files):
            want, wst = oracle_pcm(f)
            assert np.array_equal(res[k][1], want), k
    assert tm.device_parsed >= 1


def test_plan_says_which_rows_it_reads_from_the_second_form(dev, monkeypatch):
    """acmhip_plan_form_rows: whole tiles of the lean kernel for a stream that came with the form, 0 for one without, and 0 for a level-14
    stream of a plan too small for the lean kernel (its rows are read from the int16 arena)"""
    monkeypatch.delenv("ACM_K2", raising=False)
    files = [make_stream(27700, 9, 16, 12), make_stream(27701, 6, 16, 40), make_stream(27702, 14, 2, 9)]
    staged = [capi.stage_file(f) for f in files]
    ar = capi.Arena(staged)
    mf = capi.mform_streams(ar.idx, ar.descs)
    assert mf.streams[0].ntiles > 0 and mf.streams[1].ntiles == 0 and mf.streams[2].ntiles > 0
    plan = capi.Plan(dev, ar.descs, packed=mf.streams)
    assert plan.form_rows(0) == 12 * 16 and plan.form_rows(1) == 0 and plan.form_rows(2) == 0
    with pytest.raises(capi.AcmHipError):
        plan.form_rows(3)
    plan.destroy()


def test_byteplane_without_the_chunk_kernel():
    """ACM_K3=0 in a tuning build (read once per process, so this runs in a child): no level goes to the chunk kernel, every level's byte-plane tiles run
    acm_tile2's matrix build (three or four stages on the matrix cores: the one depth per level the library ships) on the form that goes
    with it - parity with the oracle over levels 7-14, even, odd and single block heights, 16-bit indices"""
    import os
    import subprocess
    import sys
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
from helpers import make_stream, oracle_pcm
from libacm_amd import capi
dev = capi.Device(0)
bad = 0
for level in range(7, 15):
    assert capi.lib().acmk_tuning_build() == 1 and capi.lib().acmhip_mform_group(level) == (16 if level >= 10 else 8)
    tr = max(capi.lib().acmhip_mform_tile_rows(level), capi.lib().acmk_tile2_rows(level), 4)
    for rows, pm in ((16, 12), (1, 9), (3, 6), (17, 15)):
        f = make_stream(28000 + level * 100 + rows, level, rows, (5 * tr + rows - 1) // rows + 1, cut=3, pwr_min=min(4, pm), pwr_max=pm,
                        val_max=65535 if pm == 15 else 255)
        got, st = capi.synth(dev, [capi.stage_file(f)], return_stats=True, mform=True)
        assert st.mform_tiles >= 5
        bad += not np.array_equal(got[0], oracle_pcm(f)[0])
print("BAD", bad)
sys.exit(1 if bad else 0)
""" % (os.path.dirname(os.path.abspath(__file__)),)
    # the shipped library reads no kernel-selection switch from the environment; its -DACM_TUNING twin (libacm_amd/lib/exp/tuning.so, built
    # by _build.build_all) does - loaded here through ACM_HIP_LIB
    from libacm_amd import _build
    env = dict(os.environ, ACM_K2="1", ACM_K3="0", ACM_HIP_LIB=_build.build_tuning())
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
