"""Host half of the product path (libacm_amd/csrc/acm_fill.cpp through acm_stage_*): the staged form
must unpack to exactly the block matrix the oracle's fill produces (reference decode.c:491-502,
586-600), statuses must follow the reference's EOF/corrupt taxonomy."""
import numpy as np
import pytest

import oracle_api as O
from helpers import golden, golden_file, make_stream
from libacm_amd import capi, synth


def unpacked(st, b):
    bl = st.block_len
    v = st.idx[b * bl:(b + 1) * bl].astype(np.int64) * int(st.hdr[b, 0])
    if st.patches is not None:
        for p in st.patches:
            if b * bl <= p.sample < (b + 1) * bl:
                v[p.sample - b * bl] = p.value
    return (v & 0xFFFFFFFF).astype(np.uint32)


def check_against_oracle_fill(data, force_chans=0):
    st = capi.stage_file(data, force_chans)
    o = O.Oracle(data, force_chans)
    for b in range(st.info.blocks):
        rc, raw, pwr, val = o.fill_next_block()
        assert rc == 1
        assert (int(st.hdr[b, 0]), int(st.hdr[b, 1])) == (val, pwr)
        assert np.array_equal(unpacked(st, b), raw.view(np.uint32)), b
    return st


@pytest.mark.parametrize("level", [0, 1, 3, 5, 7, 9, 11])
def test_staging_matrix(level):
    for rows in (1, 2, 3, 16, 17):
        for mix in (0, 1):
            st = check_against_oracle_fill(make_stream(level * 31 + rows + mix, level, rows, 3, mix=mix))
            assert st.info.blocks == 3 and st.info.end_status == 0 and st.info.npatches == 0


def test_every_filler_code():
    for case in golden()["F2_codes"]:
        st = check_against_oracle_fill(golden_file(case["file"]))
        assert st.info.blocks == 3


def test_stale_table_patches():
    """H1: indices outside [-2^pwr, 2^pwr) resolve to what earlier blocks left in the table"""
    total = 0
    for seed in range(12):
        f = make_stream(800 + seed, 5, 7, 8, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=15,
                        val_min=0, val_max=65535)
        st = check_against_oracle_fill(f)
        total += st.info.npatches
    assert total > 0


def test_statuses_follow_reference_taxonomy():
    g = golden()["F4_truncation"]
    base = golden_file(g["file"])
    for cut in g["cuts"]:
        rc, info = capi.probe(base[:cut["len"]])
        assert rc == (cut["open"] if cut["open"] < 0 else 0)
        if rc == 0:
            st = capi.stage_file(base[:cut["len"]])
            # words deliverable = what the reference delivered; a swallowed error shows as status 0 there
            assert st.words == cut["words"], cut
            if cut["words"] == 0 and cut["status"] < 0:
                assert st.info.end_status == cut["status"]
    for case in golden()["F3_corrupt"]:
        st = capi.stage_file(golden_file(case["file"]))
        assert st.words == case["words"]
        assert st.info.end_status == -6      # ACM_ERR_CORRUPT


def test_header_probe_matrix():
    for case in golden()["F6_headers"] + golden()["F5_wavc"]["bad"]:
        rc, info = capi.probe(golden_file(case["file"]))
        assert rc == (case["open"] if case["open"] < 0 else 0), case["file"]
        if rc == 0:
            want = case["info"]
            assert (info.level, info.rows, info.cols, info.channels, info.hdr_channels, info.rate) == \
                   (want["acm_level"], want["acm_rows"], want["acm_cols"], want["channels"], want["acm_channels"], want["rate"])
    g5 = golden()["F5_wavc"]
    rc, info = capi.probe(golden_file("f5_wavc"))
    assert rc == 0 and info.wavc == 1 and info.header_bytes == 42
    rc, info = capi.probe(golden_file("f5_plain"), -1)
    assert info.channels == g5["plain_quirk"]["info"]["channels"] == 2
    rc, info = capi.probe(golden_file("f5_wavc"), -1)
    assert info.channels == g5["wavc_quirk"]["info"]["channels"] == 1


def test_synth_is_deterministic_and_valid():
    a = synth.generate(seed=123, level=7, rows=16, nblocks=5)
    b = synth.generate(seed=123, level=7, rows=16, nblocks=5)
    c = synth.generate(seed=124, level=7, rows=16, nblocks=5)
    assert a == b and a != c
    pcm, st = O.Oracle.decode_all(a)
    assert st == 0 and pcm.size == 5 * 16 * 128


def test_differential_fuzz_against_oracle():
    """mutated and truncated streams: the product's parser and the oracle must agree on how many blocks
    decode, how the stream ends, and on every unpacked value (table primed so that H1 stays deterministic)"""
    rng = np.random.default_rng(99)
    bases = [bytearray(make_stream(5000 + k, lv, rows, 6, mix=k % 2, prime_table=1, channels=1 + k % 2))
             for k, (lv, rows) in enumerate([(3, 5), (5, 16), (7, 3), (2, 1), (6, 17)])]
    for it in range(400):
        g = bytearray(bases[it % len(bases)])
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(len(g) // 3, len(g)))
            g[pos] ^= 1 << int(rng.integers(0, 8))
        if it % 3 == 0:
            g = g[:int(rng.integers(len(g) // 3, len(g) + 1))]
        data = bytes(g)
        st = capi.stage_file(data)
        o = O.Oracle(data)
        b = 0
        while True:
            if b * st.block_len >= st.info.total_values:
                break
            rc, raw, pwr, val = o.fill_next_block()
            if rc != 1:
                assert b == st.info.blocks, (it, b, st.info.blocks, rc)
                assert st.info.end_status == (0 if rc == O.CLEAN_EOF else rc), (it, rc, st.info.end_status)
                break
            assert b < st.info.blocks, (it, b)
            assert np.array_equal(unpacked(st, b), raw.view(np.uint32)), (it, b)
            b += 1
        assert b == st.info.blocks
