"""acmtool (libacm_amd/csrc/acmtool.c) against transcripts of the reference tool (golden family F8):
same exit codes, stdout, stderr and output files, byte for byte (reference src/acmtool.c)."""
import base64
import os
import re
import subprocess
import tempfile

import pytest

from helpers import golden, golden_file, sha
from libacm_amd import _build

# runs that never synthesise audio (no GPU needed)
CPU_RUNS = {"version", "help", "no_command", "two_commands", "bad_option", "play_without_libao", "info", "info_mono",
            "info_stereo", "info_quiet", "decode_junk", "decode_missing", "decode_nofiles", "set_stereo", "set_mono",
            "set_on_junk", "set_on_short", "decode_o_two_files"}


def tool():
    return _build.build_tools()


def norm_err(text):
    # getopt prefixes its own diagnostics with argv[0]
    return re.sub(r"^[^\n:]*: (invalid option|option requires)", r"PROG: \1", text, flags=re.M)


def replay(rec, td):
    g = golden()["F8_cli"]
    for name, src in g["inputs"].items():
        with open(os.path.join(td, name), "wb") as f:
            f.write(golden_file(src))
    data = {"trunc.acm": golden_file("f7_src")[:len(golden_file("f7_src")) // 2],
            "junk.acm": b"this is not an acm file at all", "m.acm": golden_file("f5_plain"),
            "tiny.acm": golden_file("f5_plain")[:5]}
    for name, d in data.items():
        if name == "m.acm" and rec["label"] == "set_mono":
            # set_mono ran on the file set_stereo had already patched
            b = bytearray(d)
            b[8] = 2
            d = bytes(b)
        with open(os.path.join(td, name), "wb") as f:
            f.write(d)
    r = subprocess.run([tool()] + rec["args"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == rec["rc"], (rec["label"], r.stderr)
    want_out = base64.b64decode(rec["stdout"]) if rec["stdout_b64"] else rec["stdout"].encode("latin1")
    assert r.stdout == want_out, rec["label"]
    assert norm_err(r.stderr.decode("latin1")) == norm_err(rec["stderr"]), rec["label"]
    for name, digest in rec["files"].items():
        p = os.path.join(td, name)
        got = sha(open(p, "rb").read()) if os.path.exists(p) else None
        assert got == digest, (rec["label"], name)


def runs(pred):
    return [r for r in golden()["F8_cli"]["runs"] if pred(r["label"])]


@pytest.mark.parametrize("rec", runs(lambda l: l in CPU_RUNS), ids=lambda r: r["label"])
def test_cli_without_decoding(rec):
    with tempfile.TemporaryDirectory() as td:
        replay(rec, td)


@pytest.mark.gpu
@pytest.mark.parametrize("rec", runs(lambda l: l not in CPU_RUNS), ids=lambda r: r["label"])
def test_cli_decoding(dev, rec):
    with tempfile.TemporaryDirectory() as td:
        replay(rec, td)


@pytest.mark.gpu
def test_cli_batch_mode(dev):
    """-B: all files through one acm_batch_decode; raw output equals the per-file decode"""
    with tempfile.TemporaryDirectory() as td:
        names = []
        for k, src in enumerate(("f5_plain", "f7_src", "f1_l7_r16_c1", "f1_l9_r3_c2", "f1_l0_r3_c1")):
            p = os.path.join(td, "b%d.acm" % k)
            open(p, "wb").write(golden_file(src))
            names.append(p)
        trunc = os.path.join(td, "b_trunc.acm")
        open(trunc, "wb").write(golden_file("f7_src")[:400])
        names.append(trunc)
        for flags, ext in ((["-r"], ".raw"), ([], ".wav")):
            r = subprocess.run([tool(), "-d", "-B", "-q"] + flags + names, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr
            batch = [open(p[:-4] + ext, "rb").read() for p in names]
            berr = r.stderr
            for p in names:
                os.remove(p[:-4] + ext)
            r = subprocess.run([tool(), "-d", "-q"] + flags + names, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            single = [open(p[:-4] + ext, "rb").read() for p in names]
            assert batch == single
            assert b"adding filler_samples" in berr and berr == r.stderr


@pytest.mark.gpu
def test_c_example_batch_decode(tmp_path):
    """examples/batch_decode.c compiles as plain C99 against include/acm_hip.h and decodes what the oracle decodes"""
    import subprocess, glob
    import numpy as np
    import oracle_api as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "batch_decode")
    libdir = os.path.join(root, "libacm_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                        os.path.join(root, "examples", "batch_decode.c"), "-L", libdir, "-lacm_hip",
                        "-Wl,-rpath," + libdir, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    files = sorted(glob.glob(os.path.join(root, "tests", "golden", "acm", "f1_*.acm")))[:12]
    r = subprocess.run([exe] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files)
    for path, line in zip(files, lines):
        pcm, st = O.Oracle.decode_all(open(path, "rb").read())
        h = 2166136261
        for b in pcm.tobytes():
            h = ((h ^ b) * 16777619) & 0xFFFFFFFF
        assert line.split()[1:] == ["status", "0", "words", str(pcm.size), "fnv1a", "%08x" % h], (path, line)
