"""GPU parity: the HIP hot path (through the C ABI of libacm_hip.so) against the CPU oracle.

Bit-exact is the bar (integer path).  Reference behaviour cited: /root/reference/src/decode.c
:508-577 (juggle_block), :592-600 (amplitude table), :617-677 (write-out).
"""
import numpy as np
import pytest

from helpers import fmt_args, make_stream, oracle_pcm
from libacm_amd import capi

pytestmark = pytest.mark.gpu


def check_streams(dev, files, flags=capi.PLAN_AUTO, fmt=capi.FMT_S16LE, force_chans=0):
    staged = [capi.stage_file(f, force_chans) for f in files]
    got, st = capi.synth(dev, staged, fmt=fmt, flags=flags, return_stats=True)
    be, sg = fmt_args(fmt)
    for k, (f, g) in enumerate(zip(files, got)):
        want, status = oracle_pcm(f, force_chans, be, sg)
        assert g.size == want.size, (k, g.size, want.size)
        bad = np.nonzero(g != want)[0]
        assert bad.size == 0, "stream %d: %d/%d samples differ, first at %d (level %d rows %d)" % (
            k, bad.size, want.size, bad[0], staged[k].info.level, staged[k].info.rows)
    return st


@pytest.mark.parametrize("level", [5, 6, 7, 8, 9, 10, 11])
@pytest.mark.parametrize("rows", [1, 3, 16, 17, 64])
def test_fused_matrix(dev, level, rows):
    """fused tile kernel, every supported level x awkward row counts, several blocks (cross-block history)"""
    nblocks = max(3, (3 * (capi_tile_rows(level) - 2)) // rows + 2)     # span >= 3 tiles of the largest geometry
    nblocks = min(nblocks, 1600)
    f = make_stream(level * 100 + rows, level, rows, nblocks, cut=5)
    st = check_streams(dev, [f])
    assert st.fused_streams == 1 and st.stagewise_streams == 0


def capi_tile_rows(level):
    return (32768 if level >= 11 else 16384) >> level


@pytest.fixture
def force_k2(monkeypatch):
    """ACMHIP_PLAN_LEAN_ALWAYS: the lean tile kernel (acm_tile2) takes every whole tile, however small the plan (default: big plans only)"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | capi.PLAN_LEAN_ALWAYS)


@pytest.mark.parametrize("level", [6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("rows", [1, 3, 16, 17, 64, 700])
def test_lean_tile_kernel_matrix(dev, force_k2, level, rows):
    """acm_tile2 (whole tiles of streams decoded from row 0) + the general kernel on the ragged tail, against the oracle:
    awkward block heights (row values cross block boundaries inside a tile and inside the two rows in front of it)"""
    tr = (16384 if level >= 11 else 8192) >> level
    nblocks = max(2, (5 * tr + rows - 1) // rows + 1)
    f = make_stream(4000 + level * 100 + rows, level, rows, nblocks, cut=5)
    st = check_streams(dev, [f])
    assert st.fused_streams == 1 and st.stagewise_streams == 0 and st.launches == 2


@pytest.mark.parametrize("fmt", [capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE])
def test_lean_tile_kernel_batch(dev, force_k2, fmt):
    """many streams in one plan: workgroups start inside streams (lead-in tiles) and cross stream boundaries"""
    files = []
    for i in range(37):
        lv = 6 + i % 9
        rows = [16, 5, 33, 1][i % 4]
        files.append(make_stream(5000 + i, lv, rows, 2 + (i * 5) % 11 + ((16384 >> lv) * (1 + i % 3)) // rows,
                                 channels=1 + i % 2, cut=i % 3, val_max=65535 if i % 5 == 0 else 255, pwr_max=15 if i % 5 == 0 else 12))
    check_streams(dev, files, fmt=fmt)


@pytest.mark.parametrize("rows", [1, 3, 16, 700])
@pytest.mark.parametrize("level", [13, 14])
def test_lean_tile_kernel_levels_13_14(dev, force_k2, level, rows):
    """levels 13 / 14: four / two rows are one 128 KB tile of the lean kernel (one sixteen-wave workgroup per CU); the ragged
    tail of the stream goes through the prefix sweep + plane kernel as a window that starts inside the stream"""
    tr = 4 if level == 13 else 2
    nblocks = max(2, (5 * tr + rows - 1) // rows + 1)
    f = make_stream(4000 + level * 100 + rows, level, rows, nblocks, cut=5, channels=1 + rows % 2)
    st = check_streams(dev, [f])
    assert st.fused_streams == 1 and st.stagewise_streams == 0 and st.tiles >= 5
    g = make_stream(4242 + level, level, 4, 6)          # 24 rows = whole tiles only: nothing left for the prefix pair
    st = check_streams(dev, [g, f])
    assert st.fused_streams == 2
    for fmt in (capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE):
        check_streams(dev, [g], fmt=fmt)


def test_lean_tile_kernel_exact_multiple(dev, force_k2):
    """a stream that is a whole number of tiles leaves nothing for the general kernel"""
    f = make_stream(4242, 9, 16, 8)              # 128 rows = 8 tiles of 16 rows
    st = check_streams(dev, [f])
    assert st.launches == 1


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 7, 9, 12, 13])
@pytest.mark.parametrize("rows", [1, 2, 5, 16])
def test_stagewise_matrix(dev, level, rows):
    """generic stage-wise kernels: levels outside the fused range, and the fused levels as a cross-check"""
    f = make_stream(7000 + level * 100 + rows, level, rows, 4, cut=3)
    st = check_streams(dev, [f], flags=capi.PLAN_STAGEWISE)
    assert st.stagewise_streams == 1


@pytest.mark.parametrize("carry", ["0", "1"])
@pytest.mark.parametrize("level", [13, 14, 15])
@pytest.mark.parametrize("rows", [1, 2, 5])
def test_levels_13_to_15_prefix_plus_tile_kernel(dev, level, rows, carry, monkeypatch):
    """levels above the tile kernel's: the first level - 12 stages (acm_sw_prefix) into a scaled plane, then the level-12 tile
    kernel, halo and carry flavour, on the plane (decode.c:566-571: the later stages of level L are the cascade of level
    L - j on the same sample sequence)"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | (capi.PLAN_FORCE_CARRY if carry == "1" else capi.PLAN_FORCE_HALO))
    f = make_stream(8000 + level * 10 + rows, level, rows, 3, cut=7, val_max=65535, pwr_max=15)
    g = make_stream(8100 + level * 10 + rows, level, rows, 2, channels=2)
    st = check_streams(dev, [f, g])
    assert st.fused_streams == 2 and st.stagewise_streams == 0
    for fmt in (capi.FMT_S16BE, capi.FMT_U16LE):
        check_streams(dev, [f], fmt=fmt)


@pytest.mark.parametrize("carry", ["0", "1"])
@pytest.mark.parametrize("level,rows,blocks", [(13, 5, 30), (14, 3, 27), (15, 2, 35)])
def test_levels_13_to_15_long_streams(dev, level, rows, blocks, carry, monkeypatch):
    """more rows than one chunk of the prefix sweep (64) and than one tile of the plane kernel; windows that start inside"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | (capi.PLAN_FORCE_CARRY if carry == "1" else capi.PLAN_FORCE_HALO))
    f = make_stream(8400 + level, level, rows, blocks, cut=11)
    check_streams(dev, [f])
    s = capi.stage_file(f)
    want, _ = oracle_pcm(f)
    cols = 1 << level
    for row_begin in (1, 63, 64, 66):
        n_emit = (s.info.blocks * rows - row_begin) * cols - 11
        got = capi.synth(dev, [s], windows=[(row_begin, n_emit)])[0]
        assert np.array_equal(got, want[row_begin * cols: row_begin * cols + n_emit]), row_begin


@pytest.mark.parametrize("carry", ["0", "1"])
def test_level_13_window_and_patches(dev, carry, monkeypatch):
    """the prefix path with a window that starts inside the stream, and with H1 patches (scaled like the plane)"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | (capi.PLAN_FORCE_CARRY if carry == "1" else capi.PLAN_FORCE_HALO))
    f = make_stream(8300, 13, 2, 5)
    s = capi.stage_file(f)
    want, _ = oracle_pcm(f)
    cols = 1 << 13
    for row_begin in (1, 2, 5):
        n_emit = (s.info.blocks * 2 - row_begin) * cols - 3
        got = capi.synth(dev, [s], windows=[(row_begin, n_emit)])[0]
        assert np.array_equal(got, want[row_begin * cols: row_begin * cols + n_emit]), row_begin
    p = make_stream(8301, 13, 2, 4, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    assert capi.stage_file(p).info.npatches > 0
    check_streams(dev, [p])


@pytest.mark.parametrize("carry", ["0", "1"])
@pytest.mark.parametrize("level,rows,blocks", [(5, 4095, 3), (9, 4095, 2), (11, 4095, 2), (12, 4095, 2), (13, 4095, 2), (0, 4095, 3), (3, 4095, 2),
                                               (5, 1, 900), (9, 1, 700), (11, 1, 40), (12, 1, 24), (13, 1, 9), (2, 1, 5000),
                                               (15, 64, 2), (14, 64, 2)])
def test_header_extremes(dev, level, rows, blocks, carry, monkeypatch):
    """acm_rows is a 12-bit field (decode.c:748-750) and acm_level a 4-bit one (:747): rows 1 and 4095 through every
    kernel family the planner picks (tile kernels with and without carries, register kernel, prefix + plane kernel), and
    levels 14 / 15 with as many rows as the stress configuration has; both bit parsers"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | (capi.PLAN_FORCE_CARRY if carry == "1" else capi.PLAN_FORCE_HALO))
    f = make_stream(88000 + level * 7 + rows, level, rows, blocks, channels=1 + level % 2, cut=11)
    st = check_streams(dev, [f])
    assert st.fused_streams == 1 and st.stagewise_streams == 0
    if carry == "0":
        want, wst = oracle_pcm(f)
        for parse in (capi.PARSE_HOST, capi.PARSE_DEVICE):
            res, _ = capi.batch_decode(dev, [f], parse=parse)
            assert res[0][0] == wst and np.array_equal(res[0][1], want), parse


@pytest.mark.parametrize("form", ["int16", "byteplane"])
def test_full_size_config_1(dev, form):
    """BASELINE.json configs[1] at its full size - 1024 mono streams, level 7, 16 rows, 1000 blocks each (2.1 Gsamples) -
    through one plan, every stream's PCM compared with the CPU oracle by CRC-32 (the oracle side runs on all host cores);
    once on the int16 staged form (vector-ALU first pass), once on the byte-plane form (first pass on the matrix cores)"""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    import oracle_api as O
    from libacm_amd import workload
    threads = max(4, min(64, workload.usable_cpus()))
    b = workload.build_uniform(1024, 7, 16, 1000, keep_files=1 << 30, threads=threads)
    bufs = b.upload(dev)
    extra = ()
    try:
        if form == "byteplane":
            mf = capi.mform_streams(b.idx, b.descs, threads=threads)
            extra = mf.upload(dev)
            plan = capi.Plan(dev, b.descs, packed=mf.streams)
            plan.bind_mform(*extra)
            assert plan.stats().mform_tiles == 1024 * 1000 * 16 // 64
        else:
            plan = capi.Plan(dev, b.descs)
        dev.memset(bufs[2], 0xA5, 2 * b.pcm_words)      # a fresh allocation may hold an earlier test's PCM of the same batch (Weak 1a)
        plan.launch(*bufs)
        dev.sync()
        host = np.empty(b.pcm_words, dtype=np.uint16)
        dev.download(host, bufs[2])
        plan.destroy()
    finally:
        for p in bufs + tuple(extra):
            dev.free(p)
    raw = host.view(np.uint8)

    def one(k):
        d = b.descs[k]
        want = O.Oracle.decode_all(b.files[k].tobytes())[0]
        return zlib.crc32(want.view(np.uint8)[:2 * d.n_emit]) == zlib.crc32(raw[2 * d.pcm_off: 2 * (d.pcm_off + d.n_emit)])
    with ThreadPoolExecutor(max_workers=threads) as ex:
        ok = list(ex.map(one, range(len(b.descs))))
    assert all(ok), [k for k, v in enumerate(ok) if not v][:10]


@pytest.mark.parametrize("fmt", [capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE])
@pytest.mark.parametrize("flags", [capi.PLAN_AUTO, capi.PLAN_STAGEWISE])
def test_output_formats(dev, fmt, flags):
    """the four writers of decode.c:617-655"""
    f = make_stream(31, 7, 16, 20, cut=9, val_max=65535, pwr_max=15)
    check_streams(dev, [f], flags=flags, fmt=fmt)


def test_wraparound_values(dev):
    """large val/pwr drive the int32 arithmetic through mod-2^32 wrap and the 16-bit truncation"""
    for lv in (7, 9, 10):
        f = make_stream(77 + lv, lv, 16, 12, val_min=60000, val_max=65535, pwr_min=12, pwr_max=15, mix=1)
        check_streams(dev, [f])
        check_streams(dev, [f], flags=capi.PLAN_STAGEWISE)


def test_mixed_batch(dev):
    """one plan, streams of different level / rows / length / channels, ragged tails"""
    files = []
    for i in range(40):
        lv = [0, 3, 5, 7, 8, 9, 11, 12, 13][i % 9]
        rows = [1, 2, 16, 17, 33][i % 5] if lv < 13 else 2
        nb = 1 + (i * 7) % 9
        files.append(make_stream(500 + i, lv, rows, nb, channels=1 + i % 2, cut=i % 4))
    st = check_streams(dev, files)
    assert st.fused_streams == len(files) and st.stagewise_streams == 0      # tile kernels, register kernel, prefix + plane kernel
    st = check_streams(dev, files, flags=capi.PLAN_STAGEWISE)
    assert st.stagewise_streams == len(files)


def test_h1_stale_table_patches(dev):
    """indices outside the current table range read the previous blocks' table (hazard H1)"""
    for lv, rows in ((5, 7), (7, 16), (3, 4)):
        f = make_stream(900 + lv, lv, rows, 8, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
        s = capi.stage_file(f)
        assert s.info.npatches > 0
        check_streams(dev, [f])


def test_h1_patches_demote_tiles_not_streams(dev):
    """a few out-of-range indices in a long tile-kernel stream: only the tiles that can see a patched sample (their own
    rows and the two halo rows in front) go through the stage-wise kernels, as windows; the stream stays fused"""
    for lv, rows, nb in ((7, 16, 120), (9, 16, 40), (11, 8, 30)):
        clean = make_stream(950 + lv, lv, rows, nb, cut=11)
        dirty = make_stream(960 + lv, lv, rows, nb, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6, cut=3)
        sd = capi.stage_file(dirty)
        assert sd.info.npatches > 0
        st = check_streams(dev, [clean, dirty, clean])
        assert st.fused_streams == 3 and st.stagewise_streams == 0
        # sparse patches: keep the first, one in the middle and the last -> most tiles stay on the tile kernel; the
        # all-stage-wise plan over the same inputs is the (independent) second implementation to compare with
        keep = [0, sd.info.npatches // 2, sd.info.npatches - 1]
        got, st2 = capi.synth(dev, [sd], return_stats=True, patch_subset=keep)
        ref = capi.synth(dev, [sd], flags=capi.PLAN_STAGEWISE, patch_subset=keep)
        assert st2.fused_streams == 1 and st2.tiles > 0 and np.array_equal(got[0], ref[0])


def test_h1_patch_that_no_window_sees(dev):
    """a tile-kernel stream whose only patches lie behind what is emitted (a windowed decode that stops early) owns no
    run of the scratch plane: such patches must be dropped, not written over the plane of the stage-wise stream next to
    it (acm_hip_api.cpp, "H1 patches -> plane coordinates")"""
    dirty = make_stream(967, 7, 16, 60, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    small = make_stream(903, 3, 4, 8, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)    # level 3 + patches: stage-wise
    sd, ss = capi.stage_file(dirty), capi.stage_file(small)
    assert sd.info.npatches > 0 and ss.info.npatches > 0
    half = sd.info.blocks * sd.block_len // 2
    late = [k for k in range(sd.info.npatches) if sd.patches[k].sample >= half]
    assert late
    keep = late + [sd.info.npatches + k for k in range(ss.info.npatches)]
    n_emit = (sd.info.blocks * 16 // 4) * 128 - 5                      # the first quarter of the stream: no kept patch in sight
    windows = [(0, n_emit), (0, ss.words)]
    for order in ((sd, ss), (ss, sd)):
        w = windows if order[0] is sd else windows[::-1]
        k = keep if order[0] is sd else [ss.info.npatches + j for j in late] + list(range(ss.info.npatches))
        got, st = capi.synth(dev, list(order), windows=w, return_stats=True, patch_subset=k)
        ref = capi.synth(dev, list(order), flags=capi.PLAN_STAGEWISE, windows=w, patch_subset=k)
        assert st.fused_streams == 1 and st.stagewise_streams == 1
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        small_pcm = got[1] if order[0] is sd else got[0]
        assert np.array_equal(small_pcm, oracle_pcm(small)[0])


def test_window_with_halo(dev):
    """a window starting at row_begin > 0 needs only the two staged rows in front of it"""
    for lv, rows in ((7, 16), (9, 16), (5, 3), (11, 4), (2, 5)):
        f = make_stream(1200 + lv, lv, rows, 9)
        s = capi.stage_file(f)
        want, _ = oracle_pcm(f)
        cols = 1 << lv
        for row_begin in (1, 2, rows, 3 * rows + 1):
            n_emit = (s.info.blocks * rows - row_begin) * cols - 3
            for flags in (capi.PLAN_AUTO, capi.PLAN_STAGEWISE):
                got = capi.synth(dev, [s], flags=flags, windows=[(row_begin, n_emit)])[0]
                assert np.array_equal(got, want[row_begin * cols: row_begin * cols + n_emit]), (lv, rows, row_begin, flags)


def test_truncated_and_corrupt_streams(dev):
    """streams that end early / hit a bad filler code still decode every complete block"""
    f = make_stream(40, 7, 16, 10)
    for cut_bytes in (1, 5, 100, 1000):
        g = f[:-cut_bytes]
        check_streams(dev, [g])
    # bad filler code in the 4th block's first column: 14-byte header, find via re-synth with single code
    from libacm_amd import synth
    bad = synth.generate(seed=5, level=5, rows=4, nblocks=3, mix=synth.MIX_SINGLE, single_code=25)
    staged = capi.stage_file(bad)
    assert staged.info.blocks == 0 and staged.info.end_status == -6


def test_corpus_shaped_batch(dev):
    """BASELINE configs[2] in miniature: mixed levels 7-9, mono/stereo, ragged lengths, one plan (3 launches)"""
    from libacm_amd import workload
    shapes = workload.corpus_shapes(24, dur_min=0.05, dur_max=1.5)
    b = workload.build_corpus(len(shapes), shapes=shapes, keep_files=len(shapes))
    bufs = b.upload(dev)
    plan = capi.Plan(dev, b.descs)
    assert plan.stats().launches in (3, 6) and plan.stats().fused_streams == len(shapes)
    plan.launch(*bufs)
    out = np.zeros(b.pcm_words, dtype=np.uint16)
    dev.download(out, bufs[2])
    for d, f in zip(b.descs, b.files):
        want, st = oracle_pcm(f.tobytes())
        assert st == 0 and d.n_emit == want.size
        assert np.array_equal(out[d.pcm_off:d.pcm_off + d.n_emit], want)
    plan.destroy()
    for p in bufs:
        dev.free(p)


def test_many_short_streams_stress_shape(dev):
    """BASELINE configs[4] in miniature: thousands of 2-block stereo streams at level 11, rows 64 (every tile
    touches a stream start); a replicated stream must decode identically wherever it sits in the batch"""
    from libacm_amd import workload
    b = workload.build_uniform(96, 11, 64, 2, channels=2, keep_files=4, seed0=77)
    reps = 8
    descs = []
    per = b.descs[0].n_emit
    pad = (per + 63) // 64 * 64
    for r in range(reps):
        for d in b.descs:
            descs.append(capi.StreamDesc(idx_off=d.idx_off, hdr_off=d.hdr_off, pcm_off=(len(descs)) * pad, n_emit=d.n_emit,
                                         level=d.level, rows=d.rows, nrows=d.nrows, row_begin=0))
    d_idx, d_hdr, d_pcm0 = b.upload(dev)
    dev.free(d_pcm0)
    d_pcm = dev.malloc(len(descs) * pad * 2)
    plan = capi.Plan(dev, descs)
    plan.launch(d_idx, d_hdr, d_pcm)
    out = np.zeros(len(descs) * pad, dtype=np.uint16)
    dev.download(out, d_pcm)
    n = len(b.descs)
    for k in range(4):
        want, _ = oracle_pcm(b.files[k].tobytes())
        for r in range(reps):
            got = out[(r * n + k) * pad:(r * n + k) * pad + per]
            assert np.array_equal(got, want), (k, r)
    first = out[:n * pad]
    for r in range(1, reps):
        assert np.array_equal(out[r * n * pad:(r + 1) * n * pad], first)
    plan.destroy()
    for p in (d_idx, d_hdr, d_pcm):
        dev.free(p)


@pytest.mark.parametrize("parse", [capi.PARSE_HOST, capi.PARSE_DEVICE])
def test_batch_decode_c_api(dev, parse):
    """acm_batch_decode: threaded host parsing (or device lanes with the host reader as fallback) -> one arena -> one launch; statuses and word counts follow
    what an acm_read_loop() caller of the reference would see (truncated, corrupt, non-ACM, odd stereo)"""
    from helpers import golden, golden_file
    files = [make_stream(3100 + i, [5, 7, 9, 0, 12][i % 5], [16, 3, 1][i % 3], 2 + i % 5, channels=1 + i % 2, cut=i % 3)
             for i in range(15)]
    files += [golden_file(c["file"]) for c in golden()["F3_corrupt"]]
    files += [files[1][:len(files[1]) // 2], b"garbage", golden_file("f1_l0_r3_c2"), golden_file("f6_level0_rows1")]
    for fmt in (capi.FMT_S16LE, capi.FMT_U16BE):
        res, tm = capi.batch_decode(dev, files, fmt=fmt, threads=4, parse=parse)
        be, sg = fmt_args(fmt)
        total = 0
        if parse == capi.PARSE_DEVICE:
            assert tm.device_parsed >= 10 and tm.host_parsed >= 3      # clean streams on the GPU, broken ones not
        for (st, pcm), f in zip(res, files):
            import oracle_api as O
            o = O.Oracle(f)
            if o.err < 0:
                assert st == o.err and pcm.size == 0
                continue
            want, wst = oracle_pcm(f, 0, be, sg)
            assert np.array_equal(pcm, want), (st, wst, pcm.size, want.size)
            # the batch status is what stopped the parser; a caller looping over acm_read_loop() may see that
            # error swallowed (status 0) or a follow-up error from parsing on past it
            if st < 0:
                assert wst <= 0 and want.size < len(f) * 8
            total += want.size
        assert tm.samples == total


def test_batch_arena_is_bounded_by_the_file_not_by_its_header(dev):
    """a header may promise 2^32-1 samples; the arenas are sized by what the bytes of the file can hold, the
    item ends with its own status and the rest of the batch is untouched"""
    good = make_stream(3300, 7, 16, 6)
    want, _ = oracle_pcm(good)
    lying = bytearray(good)
    lying[4:8] = (500_000_000).to_bytes(4, "little")          # total_values (decode.c:734-738)
    lying = bytes(lying)
    tiny = lying[:19]
    assert capi.batch_pcm_words([tiny]) <= 2 * 16 * 128 + 64
    assert capi.batch_pcm_words([lying]) <= (len(lying) * 8 // (20 + 5 * 128) + 2) * 16 * 128
    for parse in (capi.PARSE_HOST, capi.PARSE_DEVICE):
        res, tm = capi.batch_decode(dev, [good, lying, tiny, good], parse=parse)
        assert res[0][0] == 0 and np.array_equal(res[0][1], want)
        assert res[3][0] == 0 and np.array_equal(res[3][1], want)
        # every block the file holds, then a clean end of stream at the block boundary (decode.c:588-589)
        assert res[1][0] == 0 and np.array_equal(res[1][1], want)
        assert res[2][1].size == 0


def test_device_parser_every_filler(dev):
    """ACM_BATCH_PARSE_DEVICE stages exactly what the host reader stages: every filler code, WAVC prefix, ragged
    rows, hazard-H1 streams (flagged -> host), and many streams at once (more lanes than one wavefront)"""
    from helpers import golden, golden_file
    g = golden()
    files = [golden_file(c["file"]) for fam in ("F1_matrix", "F2_codes", "F6_headers") for c in g[fam] if c["open"] == 0]
    files += [golden_file("f5_wavc"), golden_file("f5_plain")]
    files += [make_stream(5200 + i, [5, 6, 7, 8, 3, 10][i % 6], [16, 7, 1, 33, 512, 600][i % 6], 1 + i % 4,
                          channels=1 + i % 2, cut=i % 5) for i in range(300)]
    files += [make_stream(5900 + i, 6, 8, 3, prime_table=1, allow_out_of_range=1) for i in range(6)]
    assert len(files) > 300
    host, _ = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_HOST)
    devr, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE)
    assert tm.device_parsed > 250
    for k, ((hs, hp), (ds, dp)) in enumerate(zip(host, devr)):
        assert hs == ds and np.array_equal(hp, dp), k
    import oracle_api as O
    for k in range(0, len(files), 7):
        want, _ = oracle_pcm(files[k])
        assert np.array_equal(devr[k][1], want), k


def test_batch_parse_auto_many_streams(dev):
    """ACM_BATCH_PARSE_AUTO switches to the device parser when the batch is worth enough streams per parser thread;
    tiny streams, a few broken ones"""
    files = [make_stream(7000 + i, 5 + i % 2, 4, 1 + i % 3, channels=1 + i % 2, cut=i % 4) for i in range(2100)]
    files[17] = files[17][:30]
    files[1999] = b"nope"
    auto, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_AUTO)
    assert tm.device_parsed >= 2090 and tm.host_parsed >= 1
    few, tm2 = capi.batch_decode(dev, files[:20], threads=4, parse=capi.PARSE_AUTO)     # < 9 x threads streams' worth
    assert tm2.device_parsed == 0
    host, _ = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_HOST)
    for k, ((hs, hp), (ds, dp)) in enumerate(zip(host, auto)):
        assert hs == ds and np.array_equal(hp, dp), k
    import oracle_api as O
    for k in list(range(0, len(files), 41)) + [16, 17, 18, 1998, 1999, 2000]:     # a sample against the oracle itself
        o = O.Oracle(files[k])
        if o.err < 0:
            assert auto[k][0] == o.err and auto[k][1].size == 0, k
            continue
        want, wst = oracle_pcm(files[k])
        assert auto[k][0] == wst and np.array_equal(auto[k][1], want), k


@pytest.mark.parametrize("parse", [capi.PARSE_HOST, capi.PARSE_DEVICE])
def test_batch_pinned_output_buffers(dev, parse):
    """ACM_BATCH_PCM_PINNED: PCM is read back straight into the caller's pinned buffers, stream by stream (big streams) or
    through the arena as usual (a batch of small ones); truncated and broken files among them"""
    big = [make_stream(9300 + i, 7 + i % 3, 16, 6 + i % 5, channels=1 + i % 2, cut=i % 7) for i in range(24)]
    big[5] = big[5][:len(big[5]) // 2]
    big[11] = b"RIFFnope"
    small = [make_stream(9400 + i, 5, 4, 1 + i % 3, cut=i % 3) for i in range(40)]
    for files in (big, small, big + small):
        want, _ = capi.batch_decode(dev, files, threads=4, parse=parse)
        got, tm = capi.batch_decode(dev, files, threads=4, parse=parse, pinned=True)
        for k, ((ws, wp), (gs, gp)) in enumerate(zip(want, got)):
            assert ws == gs and np.array_equal(wp, gp), k
    for k in (0, 7, 23):
        ref, _ = oracle_pcm(big[k])
        assert np.array_equal(got[k][1], ref), k


def test_device_walk_k_columns(dev):
    """the wave-per-stream walk (acm_parse.hip: acm_parse_scan_wave) on streams that hold ONE k-filler each: every k code,
    rows around the 16-row switch of the jump table, one row, tall columns that need several 64-bit windows, columns
    per block below / at / above one wavefront's 64 offsets, truncated files"""
    from libacm_amd import synth
    files, at = [], 0
    for code in (17, 18, 20, 21, 23, 24, 26, 27):
        for rows in (1, 2, 15, 16, 17, 31, 40, 255):
            level = (3, 6, 7, 5)[at % 4]
            f = synth.generate(seed=synth.BASE_SEED + 8800 + at, level=level, rows=rows, nblocks=3, mix=synth.MIX_SINGLE,
                               single_code=code, pwr_min=12, pwr_max=12)
            files.append(f if at % 5 else f[:len(f) - 1 - at % 7])
            at += 1
    host, _ = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_HOST)
    devr, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE)
    assert tm.device_parsed >= len(files) * 3 // 4          # the truncated ones go back to the host reader
    for k, ((hs, hp), (ds, dp)) in enumerate(zip(host, devr)):
        assert hs == ds and np.array_equal(hp, dp), k
    for k in range(0, len(files), 3):
        want, _ = oracle_pcm(files[k])
        assert np.array_equal(devr[k][1], want), k


@pytest.mark.parametrize("ranges", ["2", "3", "8"])
def test_batch_block_ranges(dev, ranges, monkeypatch):
    """device parsing in block ranges (acm_batch.cpp: the walk of every stream is cut into R launches, the synthesis and the
    read-back of range r overlap the walk of range r + 1, PCM arenas range-major): forced on a small batch - streams with
    fewer blocks than ranges, ragged ends, stereo, levels with and without the lean kernel, a truncated file and streams
    with out-of-range indices (the device flags them: fix-up through the host reader), something that is not ACM"""
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    files = []
    for i in range(41):
        lv = [7, 9, 5, 3, 11, 0, 13, 8][i % 8]
        rows = [16, 3, 1, 33][i % 4]
        nb = 1 + (i * 7) % 23
        files.append(make_stream(9700 + i, lv, rows, nb, channels=1 + i % 2, cut=i % 5))
    files[6] = files[6][:len(files[6]) * 2 // 3]
    files[13] = b"RIFFnope"
    files[20] = make_stream(9790, 7, 16, 12, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    files[21] = make_stream(9791, 9, 4, 9, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6, cut=3)
    import oracle_api as O
    for rep in range(2):
        res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE)
        assert tm.host_parsed >= 2 and tm.device_parsed >= 30
        for k, f in enumerate(files):
            o = O.Oracle(f)
            if o.err < 0:
                assert res[k][0] == o.err and res[k][1].size == 0, k
                continue
            want, wst = oracle_pcm(f)
            assert res[k][0] == wst and np.array_equal(res[k][1], want), (k, ranges)


@pytest.mark.parametrize("pinned", [False, True])
def test_batch_prestaged(dev, pinned):
    """acm_batch_prestage (the host half of a batch ahead of time, no device) + acm_batch_decode on its result: same PCM and
    statuses as the oracle for clean, ragged, stereo, truncated, H1-patched and non-ACM files; items that are not the
    prestaged ones are refused"""
    files = [make_stream(9900 + i, [7, 9, 5, 3, 11, 0, 13, 8][i % 8], [16, 3, 1, 33][i % 4], 1 + (i * 7) % 19, channels=1 + i % 2, cut=i % 5)
             for i in range(29)]
    files[5] = files[5][:len(files[5]) * 2 // 3]
    files[11] = b"RIFFnope"
    files[17] = make_stream(9990, 7, 16, 12, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=6)
    res, tm = capi.batch_decode(dev, files, threads=4, pinned=pinned, prestage=True)
    import oracle_api as O
    for k, f in enumerate(files):
        o = O.Oracle(f)
        if o.err < 0:
            assert res[k][0] == o.err and res[k][1].size == 0, k
            continue
        want, wst = oracle_pcm(f)
        assert res[k][0] == wst and np.array_equal(res[k][1], want), k
    # the handle belongs to the items it was made from
    import ctypes as C
    bufs, items = capi._batch_items(files)
    pre = C.c_void_p()
    opts = capi.BatchOpts(0, capi.FMT_S16LE, 2, capi.PLAN_AUTO, capi.PARSE_HOST, 0)
    assert capi.lib().acm_batch_prestage(items, len(files), C.byref(opts), C.byref(pre), None) == 0
    try:
        opts.prestaged = pre
        assert capi.lib().acm_batch_decode(dev.h, items, len(files) - 1, C.byref(opts), None) != 0
        bufs2, items2 = capi._batch_items(files[::-1])
        assert capi.lib().acm_batch_decode(dev.h, items2, len(files), C.byref(opts), None) != 0
        # the same buffers with another length (a reused buffer): refused, not copied into arenas laid out for the new length
        items[3].len -= 40
        assert capi.lib().acm_batch_decode(dev.h, items, len(files), C.byref(opts), None) != 0
        items[3].len += 40
        assert capi.lib().acm_batch_decode(dev.h, items, len(files), C.byref(opts), None) == 0
    finally:
        capi.lib().acm_batch_prestage_free(pre)


@pytest.mark.parametrize("ranges", ["2", "4", "16"])
def test_batch_striped_upload_uneven_bit_rate(dev, ranges, monkeypatch):
    """block ranges upload the files in stripes and walk range r once stripe r + 1 is on the device: a stream that spends nearly
    all of its bits in its first blocks needs bytes that are not there yet - its walk stops (as if the data had run out) and
    the host reader takes the stream; a stream with the bits at the end never gets near the limit; neighbours are not affected"""
    from helpers import handmade_stream
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    loud, quiet = (12, 200, 13), (3, 7, 0)                          # 13-bit linear columns against empty ones
    front = handmade_stream(6, 32, [loud] * 6 + [quiet] * 18, seed=1)
    back = handmade_stream(6, 32, [quiet] * 18 + [loud] * 6, seed=2)
    even = handmade_stream(7, 16, [(9, 50, 8)] * 24, seed=3)
    files = [front, make_stream(9801, 8, 16, 9), back, even, make_stream(9802, 6, 5, 31, channels=2), front[:len(front) // 2]]
    res, tm = capi.batch_decode(dev, files, threads=4, parse=capi.PARSE_DEVICE)
    import oracle_api as O
    for k, f in enumerate(files):
        want, wst = oracle_pcm(f)
        if k == 5:
            # the batch reports what acm_read returns for the block that fails; the reference folds that into a short count when
            # the failing block is not the first of a call, so ask the oracle block by block (64 columns x 32 rows x 2 bytes)
            want, wst = O.Oracle.decode_all(f, step_bytes=2 * 32 * 64)
            want = want.view(np.uint16)
            assert wst < 0
        assert res[k][0] == wst and np.array_equal(res[k][1], want), (k, ranges, res[k][0], wst)
    if int(ranges) >= 4:
        assert tm.host_parsed >= 1              # `front` (and its truncated copy) went through the host reader
    assert tm.device_parsed >= 3


def test_device_walk_lane_kernel(dev):
    """more streams than the wave-per-stream walk takes (32 K): one stream per lane (acm_parse_scan)"""
    files = [make_stream(9100 + i % 97, 3 + i % 2, 2, 1 + i % 2, cut=i % 3) for i in range(33000)]
    devr, tm = capi.batch_decode(dev, files, threads=8, parse=capi.PARSE_DEVICE)
    assert tm.device_parsed >= 32900
    period = 582                                            # lcm(97, 2, 3): the streams repeat
    host, _ = capi.batch_decode(dev, files[:period], threads=8, parse=capi.PARSE_HOST)
    for k in range(len(files)):
        ws, wp = host[k % period]
        assert devr[k][0] == ws and np.array_equal(devr[k][1], wp), k
    for k in list(range(0, period, 7)) + [len(files) - 1, 32768, 32767]:      # a sample of them against the oracle itself
        want, wst = oracle_pcm(files[k])
        assert devr[k][0] == wst and np.array_equal(devr[k][1], want), k


@pytest.mark.parametrize("parse", [capi.PARSE_HOST, capi.PARSE_DEVICE, capi.PARSE_AUTO])
def test_batch_edge_cases(dev, parse):
    """empty batch, nothing decodable, a single stream, one parser thread, repeated calls on the same handle"""
    res, tm = capi.batch_decode(dev, [], parse=parse)
    assert res == [] and tm.samples == 0
    res, tm = capi.batch_decode(dev, [b"", b"x" * 13, b"garbage" * 10], parse=parse)
    assert all(st < 0 and pcm.size == 0 for st, pcm in res) and tm.samples == 0
    one = make_stream(8100, 7, 16, 40)
    want, _ = oracle_pcm(one)
    for threads in (1, 3):
        for _ in range(2):
            res, tm = capi.batch_decode(dev, [one], threads=threads, parse=parse)
            assert res[0][0] == 0 and np.array_equal(res[0][1], want)
    # many chunks: more streams than the 8 MiB chunk floor holds, so the pipeline really cycles
    files = [make_stream(8200 + i, 8, 16, 24, cut=i) for i in range(96)]          # 96 x 98 K samples
    res, tm = capi.batch_decode(dev, files, threads=5, parse=parse)
    for k in (0, 31, 64, 95):
        want, _ = oracle_pcm(files[k])
        assert res[k][0] == 0 and np.array_equal(res[k][1], want), k


def test_batch_device_resident_output(dev):
    """opts.d_pcm: the PCM stays in HBM at items[i].dev_off (what the multi-GPU gather consumes); too small a
    buffer is refused"""
    files = [make_stream(8300 + i, [6, 7, 9][i % 3], 16, 2 + i % 4, channels=1 + i % 2, cut=3 * i) for i in range(40)]
    files += [b"junk", files[3][:100]]
    cap = capi.batch_pcm_words(files)
    assert cap % 64 == 0 and cap > 0
    d_pcm = dev.malloc(cap * 2)
    host, _ = capi.batch_decode(dev, files, threads=4)
    for parse in (capi.PARSE_HOST, capi.PARSE_DEVICE):
        st, words, offs, tm = capi.batch_decode_device(dev, files, d_pcm, cap, threads=4, parse=parse)
        out = np.zeros(cap, dtype=np.uint16)
        dev.download(out, d_pcm)
        dev.sync()
        for k, (hs, hp) in enumerate(host):
            assert st[k] == hs and words[k] == hp.size
            assert np.array_equal(out[offs[k]:offs[k] + words[k]], hp), k
    with pytest.raises(capi.AcmHipError):
        capi.batch_decode_device(dev, files, d_pcm, cap - 64)
    dev.free(d_pcm)


@pytest.mark.parametrize("ranges", ["1", "5"])
def test_random_batch_fuzz(dev, ranges, monkeypatch):
    """200 random shapes (level 0-14, rows 1-70, 1-5 blocks, mono/stereo, ragged ends, WAVC, all four formats) in one
    batch: host parse, device parse (one walk, and block ranges) and the oracle agree stream by stream"""
    monkeypatch.setattr(capi, "BATCH_EXTRA", capi.batch_ranges(ranges))
    rng = np.random.default_rng(0xACD)
    files = []
    for i in range(200):
        level = int(rng.integers(0, 15)) if i % 10 == 0 else int(rng.integers(0, 13))
        rows = int(rng.integers(1, 71))
        nb = int(rng.integers(1, 6))
        bl = rows << level
        cut = int(rng.integers(0, min(bl, 50)))
        files.append(make_stream(9000 + i, level, rows, nb, channels=int(rng.integers(1, 3)), cut=cut,
                                 wavc=int(rng.integers(0, 4) == 0), mix=int(rng.integers(0, 2))))
    for fmt in (capi.FMT_S16LE, capi.FMT_S16BE, capi.FMT_U16LE, capi.FMT_U16BE):
        be, sg = fmt_args(fmt)
        host, _ = capi.batch_decode(dev, files, fmt=fmt, threads=4, parse=capi.PARSE_HOST)
        devr, tm = capi.batch_decode(dev, files, fmt=fmt, threads=4, parse=capi.PARSE_DEVICE)
        assert tm.device_parsed >= 190
        for k, f in enumerate(files):
            want, wst = oracle_pcm(f, 0, be, sg)
            assert np.array_equal(host[k][1], want), (k, fmt)
            assert np.array_equal(devr[k][1], want), (k, fmt)
            assert host[k][0] == devr[k][0]


@pytest.mark.parametrize("carry", ["0", "1"])
def test_tile_kernel_flavours(dev, carry, monkeypatch):
    """the halo kernel and the carry-mode kernel (ACMHIP_PLAN_FORCE_HALO / _CARRY force either) give the oracle's PCM: long streams
    (many tiles per stream, lead-in tiles where a workgroup's run starts mid-stream), ragged ends, stereo, every fused
    level, windows that start at row_begin > 0, all four output formats"""
    monkeypatch.setattr(capi, "PLAN_EXTRA", capi.PLAN_EXTRA | (capi.PLAN_FORCE_CARRY if carry == "1" else capi.PLAN_FORCE_HALO))
    files = [make_stream(9500 + lv, lv, rows, nb, channels=1 + lv % 2, cut=7 * lv)
             for lv, rows, nb in ((5, 16, 700), (6, 7, 300), (7, 16, 120), (8, 33, 40), (9, 16, 30), (10, 5, 20), (11, 64, 3))]
    for fmt in (capi.FMT_S16LE, capi.FMT_U16BE):
        be, sg = fmt_args(fmt)
        staged = [capi.stage_file(f) for f in files]
        got = capi.synth(dev, staged, fmt=fmt)
        for f, g in zip(files, got):
            want, _ = oracle_pcm(f, 0, be, sg)
            assert np.array_equal(g, want)
    for lv, rows in ((7, 16), (9, 16), (11, 4)):
        f = make_stream(9600 + lv, lv, rows, 40)
        s = capi.stage_file(f)
        want, _ = oracle_pcm(f)
        cols = 1 << lv
        for row_begin in (1, 2, rows + 1, 17 * rows):
            n_emit = (s.info.blocks * rows - row_begin) * cols - 5
            got = capi.synth(dev, [s], windows=[(row_begin, n_emit)])[0]
            assert np.array_equal(got, want[row_begin * cols: row_begin * cols + n_emit]), (lv, row_begin)


def test_concurrent_callers(dev):
    """threads sharing one device handle: acm_batch_decode calls serialise on the handle's arenas; independent
    ACMStreams of the drop-in API decode side by side on the shared default device (SURVEY 8b threading)"""
    import threading
    import oracle_api as O
    files = [make_stream(9700 + i, 5 + i % 5, 16, 6 + i % 7, channels=1 + i % 2, cut=i) for i in range(24)]
    wants = [oracle_pcm(f)[0] for f in files]
    errors = []

    def batch_worker(k):
        try:
            for rep in range(3):
                mode = (capi.PARSE_HOST, capi.PARSE_DEVICE)[(k + rep) % 2]
                res, _ = capi.batch_decode(dev, files[k::3], threads=2, parse=mode)
                for (st, pcm), want in zip(res, wants[k::3]):
                    assert st == 0 and np.array_equal(pcm, want)
        except Exception as e:      # noqa: BLE001 - collected for the main thread
            errors.append(("batch", k, repr(e)))

    def stream_worker(k):
        try:
            lib = O.bind_libacm(capi.lib())
            for f, want in list(zip(files, wants))[k::4]:
                s = O.LibacmStream(lib, f)
                pcm, rc = s.decode_all()
                s.close()
                assert pcm == want.tobytes()
        except Exception as e:      # noqa: BLE001
            errors.append(("stream", k, repr(e)))

    threads = [threading.Thread(target=batch_worker, args=(k,)) for k in range(3)]
    threads += [threading.Thread(target=stream_worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
