"""ASan + UBSan over the host half of the product (GPU sanitizers are unavailable on the pool: CPU build only).
tests/native/fuzz_host.cpp links acm_fill.cpp + acm_stream.cpp + acm_pack.cpp + acm_host_synth.cpp against stubbed device entry points and
drives mutated/truncated golden files through staging, seeks, decode-and-discard reads and reads into a buffer (host synthesis)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_parser_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_host")
    src = [os.path.join(ROOT, "tests", "native", "fuzz_host.cpp"),
           os.path.join(ROOT, "libacm_amd", "csrc", "acm_fill.cpp"),
           os.path.join(ROOT, "libacm_amd", "csrc", "acm_stream.cpp"),
           os.path.join(ROOT, "libacm_amd", "csrc", "acm_pack.cpp"),          # (the byte-plane / packed stagers: host code like the parser)
           os.path.join(ROOT, "libacm_amd", "csrc", "acm_host_synth.cpp")]    # (the host synthesis behind acm_read() where no device is used)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "libacm_amd", "csrc"), "-o", exe] + src + ["-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and "sanitize" in r.stdout and "cannot find" in r.stdout:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stdout
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "acm", "f[1257]_*.acm")))[:40]
    # ... and blocks tall and wide enough for the parser's padded scratch (rows x row bytes > 24 KB: acm_fill.cpp parse_block)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_stream
    for k, (lv, rows, nb) in enumerate(((11, 16, 3), (12, 64, 2), (9, 70, 3), (13, 5, 2))):
        path = str(tmp_path / ("tall_%d.acm" % k))
        with open(path, "wb") as f:
            f.write(make_stream(4100 + k, lv, rows, nb, channels=1 + k % 2, cut=k, pwr_max=12))
        files.append(path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, files[0], "60"] + files[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-3000:]
