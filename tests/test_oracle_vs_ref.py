"""Direct oracle-vs-reference sweeps.  Only where the compiled reference is available (authoring
container, or a box that received the prebuilt oracle/_ref); everywhere else the committed golden
vectors (test_oracle_golden.py) carry the pin."""
import numpy as np
import pytest

import oracle_api as O
from helpers import make_stream

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not present")


def both(data, **kw):
    r = O.LibacmStream(O.ref_lib(), data, kw.get("force_chans", 0))
    if r.err < 0:
        o = O.Oracle(data, kw.get("force_chans", 0))
        assert o.err == r.err
        return
    pr, sr = r.decode_all(8192, kw.get("be", 0), kw.get("sgned", 1))
    po, so = O.Oracle.decode_all(data, kw.get("force_chans", 0), be=kw.get("be", 0), sgned=kw.get("sgned", 1))
    assert so == sr and po.tobytes() == pr


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_random_matrix(level):
    for rows in (1, 2, 3, 16, 17, 64):
        if level >= 11 and rows > 17:
            continue
        for ch in (1, 2):
            for mix in (0, 1):
                both(make_stream(level * 977 + rows * 13 + ch + mix * 7, level, rows, 4, channels=ch, cut=3, mix=mix))


def test_extreme_values_and_stale_table():
    for seed in range(20):
        both(make_stream(3000 + seed, 5, 7, 8, mix=1, allow_out_of_range=1, prime_table=1, pwr_min=0, pwr_max=15,
                         val_min=0, val_max=65535))


def test_random_truncations():
    rng = np.random.default_rng(5)
    f = make_stream(99, 7, 16, 6)
    for n in rng.integers(0, len(f), size=200):
        both(f[:int(n)])


def test_bit_flips():
    rng = np.random.default_rng(6)
    # block 0 primes the whole amplitude table (pwr 15), flips stay behind it: a flipped pwr/code may then
    # index "stale" entries, but never uninitialised heap (which the reference would read as garbage)
    f = bytearray(make_stream(98, 6, 8, 6, prime_table=1))
    for _ in range(300):
        g = bytearray(f)
        pos = int(rng.integers(len(g) // 3, len(g)))
        g[pos] ^= 1 << int(rng.integers(0, 8))
        both(bytes(g))


def test_cascade_formulation_equals_reference_juggle():
    """juggle_block == `level` strided 3-tap FIR stages over the flat sample index,
    history = zeros, chunking irrelevant.  This is the formulation the HIP kernels implement."""
    P = O.refprobe_lib()
    rng = np.random.default_rng(7)
    for level in (1, 2, 3, 5, 7, 9, 10):
        for rows in (1, 3, 16, 17):
            cols = 1 << level
            nb = 4
            x = rng.integers(-2 ** 31, 2 ** 31 - 1, size=nb * rows * cols, dtype=np.int64).astype(np.int32)
            wrap = np.zeros(max(1, 2 * cols - 2), dtype=np.int32)
            ref = x.copy()
            for b in range(nb):
                blk = np.ascontiguousarray(ref[b * rows * cols:(b + 1) * rows * cols])
                P.refprobe_juggle_block(level, rows, blk.ctypes.data, wrap.ctypes.data)
                ref[b * rows * cols:(b + 1) * rows * cols] = blk
            y = x.astype(np.uint32)
            m = np.arange(y.size)
            for k in range(level):
                s = cols >> (k + 1)
                x1 = np.concatenate([np.zeros(s, np.uint32), y[:-s]]) if s < y.size else np.zeros_like(y)
                x2 = np.concatenate([np.zeros(2 * s, np.uint32), y[:-2 * s]]) if 2 * s < y.size else np.zeros_like(y)
                odd = ((m // s) & 1).astype(bool)
                y = np.where(odd, 2 * x1 - (x2 + y), 2 * x1 + (x2 + y)).astype(np.uint32)
                if k == 0:
                    y = (y + (m % (cols // 2) == 0 if cols >= 2 else 1)).astype(np.uint32)
            assert np.array_equal(y.view(np.int32), ref), (level, rows)
