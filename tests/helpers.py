"""Shared test helpers: synthetic streams + the oracle's answer for them."""
import numpy as np

import oracle_api as O
from libacm_amd import synth


def make_stream(seed, level, rows, nblocks, channels=1, cut=0, **kw):
    """One synthetic file; `cut` trims total_values so it is not block aligned."""
    total = max(1, nblocks * rows * (1 << level) - cut)
    return synth.generate(seed=synth.BASE_SEED + seed, level=level, rows=rows, nblocks=nblocks,
                          channels=channels, total_values=total, **kw)


def oracle_pcm(data, force_chans=0, be=0, sgned=1):
    """Whole-file decode by the CPU oracle -> (uint16 view of the output bytes, status)."""
    pcm, st = O.Oracle.decode_all(data, force_chans=force_chans, be=be, sgned=sgned)
    return pcm.view(np.uint16), st


def fmt_args(fmt):
    """ACMHIP_FMT_* -> (bigendianp, sgned)"""
    return (fmt & 1), (0 if fmt & 2 else 1)


def juggle_inputs(level, rows, nblocks=4):
    """Deterministic raw int32 block matrices for the juggle_block vectors (golden family F9):
    full-range values, except block 1 which has realistic magnitudes."""
    rng = np.random.default_rng([0xAC3D, level, rows])
    cols = 1 << level
    out = []
    for b in range(nblocks):
        blk = rng.integers(-2 ** 31, 2 ** 31 - 1, size=rows * cols, dtype=np.int64).astype(np.int32)
        if b == 1:
            blk = (blk >> 14).astype(np.int32)
        out.append(blk)
    return out


import hashlib
import json
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_golden = None


def golden():
    global _golden
    if _golden is None:
        with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
            _golden = json.load(f)["cases"]
    return _golden


def golden_file(name):
    with open(os.path.join(GOLDEN_DIR, "acm", name + ".acm"), "rb") as f:
        return f.read()


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def decode_record(stream_cls_factory, data, force_chans=0, be=0, sgned=1, step=8192, **io_kw):
    """Decode through any libacm.h-shaped stream wrapper and summarise like make_golden.ref_decode()."""
    s = stream_cls_factory(data, force_chans, **io_kw)
    if s.err < 0:
        return {"open": s.err}
    pcm, rc = s.decode_all(step, be, sgned)
    rec = {"open": 0, "status": rc, "words": len(pcm) // 2, "sha256": sha(pcm),
           "head": [int(x) for x in np.frombuffer(pcm[:128], dtype="<u2")],
           "tail": [int(x) for x in np.frombuffer(pcm[-128:], dtype="<u2")] if len(pcm) >= 128 else [],
           "info": s.info(), "raw_tell_end": s.getter("raw_tell")}
    s.close()
    return rec


def handmade_stream(level, rows, blocks, channels=1, rate=22050, seed=1):
    """An ACM file written by hand (bits LSB first, as libacm_amd/csrc/acm_synth.c writes them; reader: decode.c:586-589 block
    header, :491-502 column loop, :712-752 stream header): `blocks` = [(pwr, val, code)], every column of a block uses the one
    filler `code` - 0 (no payload) or a linear width 3..16 (rows x code bits, random; keep code <= pwr + 1).  For files whose
    bit rate is as uneven as one likes (the striped upload of acm_batch.cpp has to notice)."""
    rng = np.random.default_rng([0xACE5, seed, level, rows])
    acc, nbits, out = 0, 0, bytearray()

    def put(v, n):
        nonlocal acc, nbits
        acc |= (int(v) & ((1 << n) - 1)) << nbits
        nbits += n
        while nbits >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            nbits -= 8

    cols = 1 << level
    total = len(blocks) * rows * cols
    for v, n in ((0x032897, 24), (1, 8), (total & 0xFFFF, 16), (total >> 16, 16), (channels, 16), (rate, 16), (level, 4), (rows, 12)):
        put(v, n)
    for pwr, val, code in blocks:
        assert code == 0 or 3 <= code <= min(16, pwr + 1)
        put(pwr, 4)
        put(val, 16)
        for _ in range(cols):
            put(code, 5)
            for v in rng.integers(0, 1 << code, size=rows) if code else ():
                put(v, code)
    if nbits:
        put(0, 8 - nbits)
    return bytes(out)
