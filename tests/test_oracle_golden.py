"""The CPU oracle (oracle/acm_oracle.c) against the committed golden vectors, which are observations of
the compiled reference (tests/golden/make_golden.py).  Runs everywhere, no GPU, no /root/reference."""
import numpy as np
import pytest

import oracle_api as O
from helpers import golden, golden_file, juggle_inputs, sha


def oracle_record(data, force_chans=0, be=0, sgned=1, max_read=0, seekable=True):
    o = O.Oracle(data, force_chans, max_read, seekable)
    if o.err < 0:
        return {"open": o.err}
    out = []
    while True:
        rc, b = o.read(8192, be=be, sgned=sgned, loop=True)
        if rc <= 0:
            break
        out.append(b)
    pcm = b"".join(out)
    rec = {"open": 0, "status": rc, "words": len(pcm) // 2, "sha256": sha(pcm), "info": o.info(),
           "raw_tell_end": o.getter("raw_tell"),
           "head": [int(x) for x in np.frombuffer(pcm[:128], dtype="<u2")]}
    o.close()
    return rec


def same(rec, want, keys=("open", "status", "words", "sha256", "info", "raw_tell_end")):
    for k in keys:
        if k in want:
            assert rec.get(k) == want[k], (k, rec.get(k), want[k])


@pytest.mark.parametrize("family", ["F1_matrix", "F2_codes", "F3_corrupt", "F6_headers"])
def test_decode_families(family):
    for case in golden()[family]:
        same(oracle_record(golden_file(case["file"])), case)


def test_truncation_every_byte():
    g = golden()["F4_truncation"]
    base = golden_file(g["file"])
    for cut in g["cuts"]:
        rec = oracle_record(base[:cut["len"]])
        same(rec, cut, keys=("open", "status", "words", "sha256", "raw_tell_end"))


def test_wavc():
    g = golden()["F5_wavc"]
    for k in ("plain", "wavc"):
        same(oracle_record(golden_file(g[k]["file"])), g[k])
    same(oracle_record(golden_file(g["wavc_quirk"]["file"]), force_chans=-1), g["wavc_quirk"])
    same(oracle_record(golden_file(g["plain_quirk"]["file"]), force_chans=-1), g["plain_quirk"])
    assert g["plain"]["sha256"] == g["wavc"]["sha256"]
    for case in g["bad"]:
        same(oracle_record(golden_file(case["file"])), case)


def test_api_traces():
    g = golden()["F7_api"]
    src = golden_file("f7_src")
    o = O.Oracle(src)
    for step in g["reads"]:
        rc, b = o.read(step["ask"])
        assert (rc, sha(b), o.getter("pcm_tell"), o.getter("raw_tell"), o.getter("time_tell")) == \
               (step["rc"], step["sha"], step["pcm_tell"], step["raw_tell"], step["time_tell"]), step
    o.close()
    for f in g["formats"]:
        rec = oracle_record(src, be=f["be"], sgned=f["sgned"])
        assert (rec["sha256"], rec["words"]) == (f["sha256"], f["words"])
    o = O.Oracle(src)
    assert [o.read(64, wordlen=1)[0], o.read(64, wordlen=4)[0]] == g["bad_wordlen"]
    o.close()
    o = O.Oracle(src)
    for step in g["seeks"]:
        if step["op"] == "read":
            rc, b = o.read(step["arg"])
            assert (rc, sha(b), o.getter("pcm_tell")) == (step["rc"], step["sha"], step["pcm_tell"]), step
        else:
            rc = o.seek_pcm(step["arg"]) if step["op"] == "pcm" else o.seek_time(step["arg"])
            assert (rc, o.getter("pcm_tell"), o.getter("raw_tell")) == (step["rc"], step["pcm_tell"], step["raw_tell"]), step
    o.close()
    o = O.Oracle(src, seekable=False)
    o.read(256)
    assert o.seek_pcm(0) == g["noseek_back"]
    assert o.seek_pcm(300) == g["noseek_fwd"]
    o.close()
    for sr in g["short_reads"]:
        same(oracle_record(src, max_read=sr["max_read"]), sr, keys=("open", "status", "words", "sha256"))
    for e, text in g["strerror"].items():
        assert O.Oracle.strerror(int(e)) == text
    gl = g["loop_swallow"]
    o = O.Oracle(golden_file(gl["file"]))
    rc1, b1 = o.read(4096, loop=True)
    rc2, _ = o.read(4096, loop=True)
    rc3, _ = o.read(4096, loop=True)
    assert [rc1, rc2, rc3] == gl["rc"] and sha(b1) == gl["sha"]


def test_getters():
    g = golden()["F7_api"]["getters"]
    src = golden_file("f7_src")
    for label, fc in (("plain", 0), ("force1", 1), ("force2", 2)):
        o = O.Oracle(src, fc)
        for k, v in g[label].items():
            if k == "info":
                assert o.info() == v
            else:
                assert o.getter(k) == v, (label, k)
        o.close()


def test_juggle_and_output_vectors():
    """juggle_block + the four writers on raw block matrices (reference decode.c:528-577, 617-677)"""
    for rec in golden()["F9_juggle"]:
        level, rows = rec["level"], rec["rows"]
        cols = 1 << level
        wrap = np.zeros(max(1, 2 * cols - 2), dtype=np.int32)
        last = None
        for blk, want in zip(juggle_inputs(level, rows), rec["blocks"]):
            blk = blk.copy()
            O.Oracle.juggle_block(level, rows, blk, wrap)
            assert sha(blk.tobytes()) == want["sha256"], (level, rows)
            assert [int(x) for x in blk[:8]] == want["head"]
            last = blk
        assert sha(wrap.tobytes()) == rec["wrap_sha256"]
        for be in (0, 1):
            for sg in (0, 1):
                rc, dst = O.Oracle.output(last, level, be, sg)
                assert rc == 2 * last.size
                assert sha(dst.tobytes()) == rec["pcm"]["be%d_s%d" % (be, sg)]
