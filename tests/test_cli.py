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


def replay(rec, td, env=None):
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
    r = subprocess.run([tool()] + rec["args"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
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
    """the decoding transcripts with every stream on the GPU (ACMTOOL_HOST_LIMIT=0: by default streams this short stay on the host)"""
    with tempfile.TemporaryDirectory() as td:
        replay(rec, td, env={"ACMTOOL_HOST_LIMIT": "0"})


@pytest.mark.parametrize("rec", runs(lambda l: l not in CPU_RUNS), ids=lambda r: r["label"])
def test_cli_decoding_on_the_host(rec):
    """the same transcripts with the tool's defaults: short streams are synthesised on the host (acm_host_synth.cpp) - and on a box
    without a GPU, like the authoring container, everything is (BASELINE.json configs[0])"""
    with tempfile.TemporaryDirectory() as td:
        replay(rec, td)


@pytest.mark.gpu
@pytest.mark.parametrize("detach", ["1", "0"], ids=["detached", "one_process"])
def test_cli_returns_when_the_outputs_are_closed(dev, detach):
    """ACMTOOL_DETACH=1: acmtool decodes in a child that reports its exit code once every output is written and closed (the
    kernel's teardown of the GPU process is then not the caller's to wait for); the default is one process.  Either way: the
    same files, the same messages, the same exit code - and the files are complete the moment the tool returns (with pipes
    for stdout / stderr too: the detached child closes its ends before it goes on to die)."""
    env = dict(os.environ, ACMTOOL_DETACH=detach)
    with tempfile.TemporaryDirectory() as td:
        from helpers import make_stream
        files = {"a.acm": make_stream(7700, 7, 16, 40), "b.acm": make_stream(7701, 9, 4, 11, channels=2), "c.acm": b"not an acm file"}
        for n, d in files.items():
            open(os.path.join(td, n), "wb").write(d)
        for flags in (["-d", "-q", "-r"], ["-d", "-q", "-B"]):
            r = subprocess.run([tool()] + flags + sorted(files), cwd=td, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            ext = ".raw" if "-r" in flags else ".wav"
            import oracle_api as O
            for n in ("a", "b"):
                got = open(os.path.join(td, n + ext), "rb").read()          # complete right now
                want = O.Oracle.decode_all(files[n + ".acm"])[0].tobytes()
                assert got[-len(want):] == want and len(got) == len(want) + (0 if ext == ".raw" else 44), (flags, n)
                os.remove(os.path.join(td, n + ext))
            assert not os.path.exists(os.path.join(td, "c" + ext))
            assert r.returncode == 0 and b"c.acm" in r.stderr, flags
        # a child that does not get as far as reporting: the parent passes its status on
        rc = subprocess.call([tool(), "-d", "-q", "-o", os.path.join(td, "nodir", "x.wav"), "a.acm"], cwd=td, env=env,
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert rc == subprocess.call([tool(), "-d", "-q", "-o", os.path.join(td, "nodir", "x.wav"), "a.acm"], cwd=td,
                                     env=dict(env, ACMTOOL_DETACH="0"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.mark.gpu
def test_cli_detached_child_follows_its_parent(dev):
    """a signal that ends the process the caller waits for must not leave the GPU child decoding as an orphan.
    A long decode, detached; SIGTERM to the waiting process: it reports 128 + SIGTERM and the child is gone with it (the output
    file stops growing).  Under a tool library that initialises the GPU before main() (rocprofv3's) the tool never forks."""
    import signal
    import time
    from helpers import make_stream
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "long.acm"), "wb").write(make_stream(7710, 9, 16, 6000))          # 49 Msamples
        env = dict(os.environ, ACMTOOL_DETACH="1")
        p = subprocess.Popen([tool(), "-d", "-q", "-r", "long.acm"], cwd=td, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        time.sleep(0.05)
        kids = [int(x) for x in subprocess.run(["pgrep", "-P", str(p.pid)], stdout=subprocess.PIPE, text=True).stdout.split()]
        p.send_signal(signal.SIGTERM)
        rc = p.wait(30)
        if kids:                                    # it had forked already: the child got the signal too
            assert rc in (128 + signal.SIGTERM, -signal.SIGTERM), rc
            deadline = time.time() + 10
            while time.time() < deadline and os.path.exists("/proc/%d" % kids[0]) and \
                    open("/proc/%d/stat" % kids[0]).read().split(")")[1].split()[0] != "Z":
                time.sleep(0.05)
            assert not os.path.exists("/proc/%d" % kids[0]) or open("/proc/%d/stat" % kids[0]).read().split(")")[1].split()[0] == "Z"
        # with a tool library announced in the environment: one process, same result
        # (a preload that does not exist: ld.so says so on stderr and goes on; ROCP_TOOL_LIBRARIES would make the runtime look for it)
        env2 = dict(env, LD_PRELOAD="/nonexistent/librocprofiler-sdk-tool.so")
        open(os.path.join(td, "s.acm"), "wb").write(make_stream(7711, 7, 16, 40))
        p = subprocess.Popen([tool(), "-d", "-q", "-r", "s.acm"], cwd=td, env=env2, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        time.sleep(0.02)
        assert not subprocess.run(["pgrep", "-P", str(p.pid)], stdout=subprocess.PIPE, text=True).stdout.split()
        assert p.wait(60) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("budget", [None, "20000"], ids=["one_group", "many_groups"])
def test_cli_batch_mode(dev, budget):
    """-B: groups of files through acm_batch_decode with reading / decoding / writing overlapped.  The files it leaves
    behind and its stderr are the REFERENCE tool's from a one-by-one decode of the same files (tests/golden/f8_batch.json,
    recorded from oracle/_ref/acmtool_ref by tests/golden/make_golden_batch.py), whatever the group size."""
    import json
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f8_batch.json")))
    env = dict(os.environ)
    if budget:
        env["ACMTOOL_BATCH_BYTES"] = budget         # a handful of files per group: several groups in flight
    with tempfile.TemporaryDirectory() as td:
        names = []
        for name, src, cut in g["inputs"]:
            data = b"this is not an acm file at all" if src is None else golden_file(src)
            with open(os.path.join(td, name), "wb") as f:
                f.write(data if cut is None else data[:cut])
            names.append(name)
        for run in g["runs"]:
            for out in run["files"]:
                if os.path.exists(os.path.join(td, out)):
                    os.remove(os.path.join(td, out))
            r = subprocess.run([tool(), "-d", "-B", "-q"] + run["flags"] + names, cwd=td, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == run["rc"], r.stderr
            assert r.stderr.decode("latin1") == run["stderr"], run["flags"]
            assert r.stdout.decode("latin1") == run["stdout"]
            for out, digest in run["files"].items():
                p = os.path.join(td, out)
                got = sha(open(p, "rb").read()) if os.path.exists(p) else None
                assert got == digest, (run["flags"], out)


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
