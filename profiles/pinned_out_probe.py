#!/usr/bin/env python3
"""Repro harness for VERDICT r3 Weak 5 (device-parse batch, pinned caller buffers: read-back sometimes 2x slower).
Runs the batch again and again, pageable and pinned output alternating, and after every call prints where the pages of every
mapping of a gigabyte or more live (/proc/self/numa_maps: N0= / N1= page counts, kernelpagesize).  GPU box only.
usage: python3 profiles/pinned_out_probe.py [runs]"""
import os
import re
import sys

sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from libacm_amd import capi, synth


def big_mappings():
    out = []
    with open("/proc/self/numa_maps") as f:
        for line in f:
            parts = line.split()
            pages = {k: int(v) for k, v in (p.split("=") for p in parts[2:] if re.fullmatch(r"N\d+=\d+", p))}
            kps = [int(p.split("=")[1]) for p in parts if p.startswith("kernelpagesize_kB=")]
            total = sum(pages.values()) * (kps[0] if kps else 4)
            if total >= 1 << 20:        # kB
                out.append("%s:%s" % (parts[0][-9:], ",".join("%s=%.1fG" % (k, v * (kps[0] if kps else 4) / (1 << 20)) for k, v in sorted(pages.items()))))
    return " ".join(out)


runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ.setdefault("ACM_BATCH_RANGES", "16")
dev = capi.Device(0)
with ThreadPoolExecutor(32) as ex:
    files = list(ex.map(lambda i: synth.generate(seed=synth.BASE_SEED + i, level=9, rows=16, nblocks=250), range(1024)))
for rep in range(runs):
    for pinned in (False, True):
        res, tm = capi.batch_decode(dev, files, parse=capi.PARSE_DEVICE, pinned=pinned)
        del res
        print("%-10s total %.4f s (stage %.3f h2d %.3f d2h %.3f) | %s" % ("pinned-out" if pinned else "pageable", tm.total_s, tm.stage_s, tm.h2d_s, tm.d2h_s, big_mappings()), flush=True)
