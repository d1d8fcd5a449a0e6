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
