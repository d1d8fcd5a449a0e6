#!/usr/bin/env python3
"""Kernel-only throughput of every built-in tile-kernel variant at every fused level (tuning aid).
usage (GPU box): python3 profiles/sweep_variants.py [streams]"""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, json, os
sys.path.insert(0, %r)
from libacm_amd import capi, workload
dev = capi.Device(0)
out = {}
for level, rows, blocks in ((5, 16, 4000), (6, 16, 2000), (7, 16, 1000), (8, 16, 500), (9, 16, 250), (10, 16, 125), (11, 64, 16), (12, 64, 8)):
    b = workload.build_uniform(int(sys.argv[1]), level, rows, blocks, seed0=level << 12)
    bufs = b.upload(dev)
    plan = capi.Plan(dev, b.descs)
    for _ in range(2):
        plan.launch(*bufs)
    ms = plan.time(*bufs, reps=6) / 6
    out[level] = round(b.samples / ms / 1e6, 1)       # Gsamples/s
    plan.destroy()
    for p in bufs:
        dev.free(p)
print(json.dumps(out))
''' % ROOT

streams = sys.argv[1] if len(sys.argv) > 1 else "512"
nvar = int(os.environ.get("NVAR", "5"))
variants = [int(x) for x in os.environ["VARIANTS"].split(",")] if os.environ.get("VARIANTS") else list(range(nvar))
rows = {}
for v in variants:
    env = dict(os.environ, ACM_K1_VARIANT=str(v))
    r = subprocess.run([sys.executable, "-c", CODE, streams], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        rows[v] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        rows[v] = {}
levels = [5, 6, 7, 8, 9, 10, 11, 12]
print("Gsamples/s   " + "".join("L%-8d" % l for l in levels))
for v, d in rows.items():
    print("variant %d    " % v + "".join("%-9s" % d.get(str(l), "-") for l in levels))
best = {l: max(rows, key=lambda v: rows[v].get(str(l), 0)) for l in levels}
print("best         " + "".join("v%-8d" % best[l] for l in levels))
