"""Generated-code invariants of the lean tile kernels (libacm_amd/csrc/acm_kernels.hip: acm_tile2, its packed-form build
acm_tile2p - their mangled names both match `acm_tile2` - and the chunk kernel acm_chunk, which issues and counts its vector memory
operations the same way).

acm_tile2 issues its global loads from inline asm and waits for them by hand (one `s_waitcnt vmcnt(N)` at the end
of the iteration that issued them), so the compiler does not know that the destination registers of those loads are
not valid until that wait.  It must therefore not READ such a register (copy it, spill it, use it) between the load
and the next `s_waitcnt vmcnt`.  This test compiles the kernels to gfx950 assembly (hipcc cross-compiles without a
GPU) and checks exactly that, plus that no compiler-tracked vector load crept into the kernel (its automatic waits
would be computed without the asm loads)."""
import os
import re
import subprocess

import pytest

from libacm_amd import _build


def regs_of(tok):
    """'v12' -> {12}; 'v[4:7]' -> {4,5,6,7}; anything else -> empty"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(line):
    body = line.split(";")[0].strip()
    if not body or body.endswith(":") or body.startswith("."):
        return None, []
    parts = body.split(None, 1)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    ops = [re.sub(r"^(sext|neg|abs)\((.*)\)$", r"\2", o.split(" ")[0]) for o in ops]
    return parts[0], ops


@pytest.fixture(scope="module")
def kernel_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "acm_kernels.s"
    cmd = [_build.HIPCC, "-O3", "-std=c++17", "--offload-arch=" + _build.GFX, "-I", _build.INC, "-I", _build.CSRC,
           "--cuda-device-only", "-S", "-o", str(out), os.path.join(_build.CSRC, "acm_kernels.hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    return out.read_text()


def tile2_bodies(asm):
    for m in re.finditer(r"^(_ZN\S*(?:acm_tile2|acm_chunk)\S*):", asm, re.M):
        end = asm.index(".Lfunc_end", m.end())
        yield m.group(1), asm[m.end():end].split("\n")


def basic_blocks(lines):
    """[(label or None, [(line_no, op, ops, text)], successors as labels / 'fall')] of one function body"""
    blocks, cur, label = [], [], None
    for ln, line in enumerate(lines):
        t = line.split(";")[0].strip()
        m = re.fullmatch(r"(\.L\w+):", t)         # the compiler's blocks and the skip labels inside the second-row load asm
        if m:
            blocks.append([label, cur, None])
            cur, label = [], m.group(1)
            continue
        op, ops = operands(line)
        if op is None:
            continue
        cur.append((ln, op, ops, line.strip()))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            blocks.append([label, cur, None])
            cur, label = [], None
    blocks.append([label, cur, None])
    index = {b[0]: k for k, b in enumerate(blocks) if b[0]}
    for k, b in enumerate(blocks):
        succ = []
        last = b[1][-1] if b[1] else None
        if last and last[1] == "s_branch":
            succ = [index[last[2][0]]]
        elif last and last[1].startswith("s_cbranch"):
            succ = [index[last[2][0]]] + ([k + 1] if k + 1 < len(blocks) else [])
        elif last and last[1] in ("s_endpgm", "s_setpc_b64"):
            succ = []
        elif k + 1 < len(blocks):
            succ = [k + 1]
        b[2] = succ
    return blocks


def reads_of(op, ops, text):
    srcs = ops if op.startswith(("global_store", "ds_write", "s_")) else ops[1:]
    read = set()
    for o in srcs:
        read |= regs_of(o)
    if op.startswith("v_") and ops and "UNUSED_PRESERVE" in text:
        read |= regs_of(ops[0])                                  # SDWA that keeps the rest of its destination
    return read


def test_no_read_of_a_loading_register_before_the_wait(kernel_asm):
    """forward data flow over the kernel's control-flow graph: `pending` = registers with a hand-issued load in flight
    (set by a global_load, cleared by any s_waitcnt vmcnt); no instruction may read a pending register on any path"""
    n_kernels = 0
    for name, lines in tile2_bodies(kernel_asm):
        n_kernels += 1
        blocks = basic_blocks(lines)
        state_in = [set() for _ in blocks]
        work = list(range(len(blocks)))
        n_loads = sum(1 for b in blocks for ins in b[1] if ins[1].startswith("global_load"))
        while work:
            k = work.pop()
            pending = set(state_in[k])
            for ln, op, ops, text in blocks[k][1]:
                if op == "s_waitcnt" and "vmcnt" in text:
                    pending = set()
                    continue
                bad = reads_of(op, ops, text) & pending
                assert not bad, "%s line %d reads v%s while its load is in flight: %s" % (name[:60], ln, sorted(bad), text)
                # ... nor WRITE one: a loaded value the compiler finds dead (round 4 met the first positions of a warm-up body once
                # their multiplies were single statements) leaves its register free for anything else while the load is still on its way
                if not op.startswith(("global_store", "ds_write", "s_", "buffer_store")) and ops:
                    clobbered = regs_of(ops[0]) & pending
                    assert not clobbered, "%s line %d writes v%s while a load into it is in flight: %s" % (name[:60], ln, sorted(clobbered), text)
                if op.startswith("global_load"):
                    pending |= regs_of(ops[0])
            for t in blocks[k][2]:
                if not pending <= state_in[t]:
                    state_in[t] |= pending
                    work.append(t)
        # prologue + in-loop sets of staged-index loads (the matrix-core builds: 8 + the row values; the chunk kernel asks for as few as four
        # in its loop where the rows in front of a walk stay in registers)
        # (and the six-stage first pass inside acm_tile2 - levels 13, 14 -: eight loads for the four rows of a run's first tile, four per tile after it)
        assert n_loads >= (12 if "acm_chunk" in name or "Lb1ELi6E" in name else 2 * 9), (name, n_loads)
    assert n_kernels >= 26                                 # nine levels of acm_tile2, its matrix build at eight (one depth each), five levels of the chunk kernel, four of acm_tile2p


def test_every_wait_is_written_by_hand(kernel_asm):
    """exactly two vmcnt waits per kernel: after the prologue loads (0) and, unconditionally, at the end of every
    iteration behind its PCM stores (their number; a lead-in tile stores into a sink)"""
    for name, lines in tile2_bodies(kernel_asm):
        assert not any(l.strip().startswith("scratch_") for l in lines), name[:60]     # a spill is a compiler-tracked vector access
        stores = [l for l in lines if l.strip().startswith("global_store")]
        assert len(stores) in (4, 8), (name[:60], len(stores))
        if ("Lb1E" in name and "acm_tile2I" in name) or "acm_chunk" in name:
            # the matrix-core builds first fill their coefficient tables from constant memory: compiler-tracked loads and their waits,
            # all in front of the first hand-issued load (nothing of the kernel's own is in flight there)
            first_hand = next(k for k, l in enumerate(lines) if l.strip().startswith(";;#ASMSTART") and lines[k + 1].strip().startswith("global_load"))
            lines = lines[first_hand:]
        waits = [l.strip() for l in lines if "vmcnt" in l]
        assert sorted(waits) == ["s_waitcnt vmcnt(0)", "s_waitcnt vmcnt(%d)" % len(stores)], (name[:60], waits)


def test_parse_walk_owns_m0(tmp_path):
    """acm_parse_scan_wave routes a v_writelane lane select through m0 by hand (acm_parse.hip); nothing the compiler
    generated in that kernel may touch m0, and the kernel must stay free of scratch"""
    out = tmp_path / "acm_parse.s"
    cmd = [_build.HIPCC, "-O3", "-std=c++17", "--offload-arch=" + _build.GFX, "-I", _build.INC, "-I", _build.CSRC,
           "--cuda-device-only", "-S", "-o", str(out), os.path.join(_build.CSRC, "acm_parse.hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    asm = out.read_text()
    m = re.search(r"^(_ZN\S*acm_parse_scan_wave\S*):", asm, re.M)
    assert m
    body = asm[m.end():asm.index(".Lfunc_end", m.end())].split("\n")
    by_hand, mine, writes = False, 0, 0
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            by_hand = True
        elif t.startswith(";;#ASMEND"):
            by_hand = False
        elif t and not t.startswith(";"):
            code = t.split(";")[0]
            if by_hand:
                mine += "m0" in code
                writes += code.startswith("v_writelane_b32")
            else:
                assert not re.search(r"\bm0\b", code), t
                assert not t.startswith("scratch_"), t
    assert writes >= 1 and mine == 2 * writes
    # ... and so must the column kernel: its row loop exists twice (plain blocks / blocks whose first or last row pair has another class);
    # written as a generic lambda the two copies kept their captures in scratch memory and the kernel ran 8 x slower (profiles/r6_level9_notes.txt 17)
    m = re.search(r"^(_ZN\S*acm_parse_columns\S*):", asm, re.M)
    assert m
    cols = asm[m.end():asm.index(".Lfunc_end", m.end())]
    assert "scratch_" not in cols and cols.count("global_store_byte") >= 4
    for kernel in ("acm_parse_columns", "acm_parse_scan_wave"):
        md = re.search(r"\.amdhsa_kernel \S*" + kernel + r"\S*\n(?:.*\n)*?\s*\.amdhsa_private_segment_fixed_size (\d+)", asm)
        assert md and md.group(1) == "0", kernel


def test_phase_priorities_are_in_the_tile_loop(kernel_asm):
    """acm_tile2 raises the wave priority for the LDS passes above the first pass (acm_kernels.hip: phase_prio): every
    build with >= 2 workgroups per CU sets three different levels per iteration; the 1024-thread builds of levels 13 and 14 (one
    sixteen-wave workgroup per CU: every wave of a SIMD is in the same phase) set none"""
    n = 0
    for name, lines in tile2_bodies(kernel_asm):
        prios = [l.split()[1] for l in lines if l.strip().startswith("s_setprio")]
        if "TileCfgILi13ELi1024E" in name or "TileCfgILi14ELi1024E" in name:
            assert prios == [], (name[:60], prios)
        else:
            assert set(prios) == {"0", "2", "3"}, (name[:60], prios)
        n += 1
    assert n >= 13


def test_chunk_kernel_keeps_its_tables_in_lds(kernel_asm):
    """acm_chunk: no flat access (a table pointer picked at run time loses its address space, and a flat load waits for every vector memory
    operation in flight), no scratch, one workgroup barrier (behind the table fill) and none in the loop, the matrix instruction it is
    written around"""
    n = 0
    for name, lines in tile2_bodies(kernel_asm):
        if "acm_chunk" not in name:
            continue
        n += 1
        text = "\n".join(l.split(";")[0] for l in lines)
        assert "flat_" not in text and "scratch_" not in text and "buffer_load" not in text, name[:60]
        assert text.count("s_barrier") == 1, name[:60]
        assert text.count("v_mfma_i32_16x16x64_i8") >= 18 and "v_mul_lo_u32" not in text, name[:60]
    assert n >= 1


def test_packed_build_reads_its_descriptors_through_the_scalar_cache(kernel_asm):
    """acm_tile2p: chunk descriptors and tile records are scalar loads (a vector load would bring a compiler-placed vmcnt wait
    with it: checked above), the chunk data are the hand-issued dword loads inside the branchy asm statements"""
    n = 0
    for name, lines in tile2_bodies(kernel_asm):
        if "acm_tile2p" not in name:
            continue
        n += 1
        text = "\n".join(lines)
        assert "flat_load" not in text and "buffer_load" not in text, name[:60]
        assert text.count("s_load_dwordx8") >= 2, name[:60]
        assert text.count(".Lacm_pk") >= 10 and text.count(".Lacm_un_") >= 10, name[:60]
    assert n == 4
