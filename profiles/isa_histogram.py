#!/usr/bin/env python3
"""Instruction histogram of one acm_tile2 instantiation, region by region (regions = code between s_barriers), from the
gfx950 assembly hipcc produces (cross-compiles without a GPU).  usage: python3 profiles/isa_histogram.py [level]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
level = int(sys.argv[1]) if len(sys.argv) > 1 else 9
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-I",
                    os.path.join(ROOT, "libacm_amd", "csrc"), "--cuda-device-only", "-S", "-o", out,
                    os.path.join(ROOT, "libacm_amd", "csrc", "acm_kernels.hip")], check=True, stderr=subprocess.DEVNULL)
    asm = open(out).read()
m = re.search(r"^(_ZN\S*acm_tile2INS_7TileCfgILi%dE\S*):" % level, asm, re.M)
body = asm[m.end():asm.index(".Lfunc_end", m.end())]
tail = asm[asm.index(".Lfunc_end", m.end()):][:3000]
print("# %s" % m.group(1)[:110])
for key in ("NumVgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
    print("# %s %s" % (key, re.search(r"; %s: (\d+)" % key, tail).group(1)))


def klass(op):
    if op in ("v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_xor_b32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_mov_b32_e32",
              "v_lshrrev_b32_e32", "v_ashrrev_i32_e32"):
        return "VALU simple (2.25 cyc)"
    if op.startswith("v_"):
        return "VALU other (4.2 cyc)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "scratch_", "buffer_")):
        return "VMEM"
    if op in ("s_waitcnt", "s_nop", "s_barrier"):
        return op
    return "SALU/branch"


regions, cur = [], collections.Counter()
ops_all = collections.Counter()
for line in body.split("\n"):
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        continue
    op = t.split()[0]
    ops_all[op] += 1
    cur[klass(op)] += 1
    if op == "s_barrier":
        regions.append(cur)
        cur = collections.Counter()
regions.append(cur)
names = ["prologue (first tile's loads, loop entry) .. top barrier"]
names += ["first pass + issue of the next tile's loads + LDS pass 1 warm-up reads"]
k = 1
while len(names) < len(regions) - 1:
    names += ["LDS pass %d: carry save, body reads .. pass %d warm-up reads" % (k, k + 1)] if len(names) % 2 == 0 else ["(between the two barriers of a pass)"]
    k += len(names) % 2 == 0
names += ["write-out (LDS gather, PCM stores), counted wait, loop control"]
cols = ["VALU simple (2.25 cyc)", "VALU other (4.2 cyc)", "LDS", "VMEM", "SALU/branch", "s_waitcnt", "s_nop"]
print("%-4s %s" % ("reg", "  ".join("%-22s" % c for c in cols)))
for i, r in enumerate(regions):
    print("%-4d %s" % (i, "  ".join("%-22d" % r[c] for c in cols)))
tot = sum(regions, collections.Counter())
print("%-4s %s" % ("sum", "  ".join("%-22d" % tot[c] for c in cols)))
print("# static counts of straight-line code; both write-out formats of the last pass are present (one executes per launch)")
print("# most frequent opcodes: " + ", ".join("%s %d" % kv for kv in ops_all.most_common(14)))
