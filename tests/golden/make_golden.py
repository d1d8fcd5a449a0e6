#!/usr/bin/env python3
"""Regenerate tests/golden/ from the REAL reference (markokr/libacm v1.3).

Runs only in the authoring container: it needs oracle/_ref (built by `make -C oracle ref`
from /root/reference/src).  Inputs are synthetic ACM files written by our own synthesiser
(libacm_amd/csrc/acm_synth.c) or hand-assembled bit strings below; every expected value in
golden.json / juggle.npz is an observation of the compiled reference, never of our code.

  python tests/golden/make_golden.py

Families: F1 level x rows matrix, F2 one stream per filler code, F3 corrupt
codes, F4 truncation at every byte, F5 WAVC, F6 header validation, F7 API traces (read sizes,
seeks, output formats, short reads), F8 acmtool transcripts, F9 raw juggle_block vectors.
"""
import base64
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as O  # noqa: E402
from libacm_amd import synth  # noqa: E402

ACM_DIR = os.path.join(HERE, "acm")


def sha(b):
    return hashlib.sha256(b).hexdigest()


class BitWriter:
    """LSB-first bit packer for the hand-made cases."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, bits):
        self.acc |= (v & ((1 << bits) - 1)) << self.n
        self.n += bits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def done(self):
        if self.n:
            self.put(0, 8 - self.n)
        return bytes(self.out)


def header(w, total, channels=1, rate=22050, level=3, rows=4, magic=0x032897, version=1):
    w.put(magic, 24)
    w.put(version, 8)
    w.put(total & 0xFFFF, 16)
    w.put(total >> 16, 16)
    w.put(channels, 16)
    w.put(rate, 16)
    w.put(level, 4)
    w.put(rows, 12)


def ref_decode(data, force_chans=0, be=0, sgned=1, step=8192, **io_kw):
    s = O.LibacmStream(O.ref_lib(), data, force_chans, **io_kw)
    if s.err < 0:
        return {"open": s.err}
    pcm, rc = s.decode_all(step, be, sgned)
    info = s.info()
    rec = {"open": 0, "status": rc, "words": len(pcm) // 2, "sha256": sha(pcm),
           "head": [int(x) for x in np.frombuffer(pcm[:128], dtype="<u2")],
           "tail": [int(x) for x in np.frombuffer(pcm[-128:], dtype="<u2")] if len(pcm) >= 128 else [],
           "info": info, "total_values": s.getter("pcm_total") * info["channels"],
           "raw_tell_end": s.getter("raw_tell")}
    s.close()
    return rec


def save(name, data):
    with open(os.path.join(ACM_DIR, name + ".acm"), "wb") as f:
        f.write(data)


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(ACM_DIR, exist_ok=True)
    for f in os.listdir(ACM_DIR):
        os.remove(os.path.join(ACM_DIR, f))
    G = {"reference": "markokr/libacm v1.3, compiled by oracle/Makefile `ref`", "cases": {}}
    C = G["cases"]

    # ---- F1: level x rows matrix, >= 3 blocks, total_values not block aligned, mono + stereo
    f1 = []
    for level in (0, 1, 2, 5, 7, 9, 11):
        for rows in (1, 3, 16, 17, 64):
            if (level >= 9 and rows > 16) or (level == 11 and rows > 3):
                continue
            for ch in (1, 2):
                nb = 3 if level < 9 else 4
                name = "f1_l%d_r%d_c%d" % (level, rows, ch)
                total = max(1, nb * rows * (1 << level) - 5)
                d = synth.generate(seed=synth.BASE_SEED + level * 1000 + rows * 10 + ch, level=level, rows=rows,
                                   nblocks=nb, channels=ch, total_values=total, pwr_min=3, pwr_max=8)
                save(name, d)
                f1.append({"file": name, **ref_decode(d)})
    C["F1_matrix"] = f1

    # ---- F2: every valid filler code on its own, odd and even row counts (last-row early breaks)
    f2 = []
    for code in synth.VALID_CODES:
        for rows in (5, 6):
            name = "f2_code%d_r%d" % (code, rows)
            d = synth.generate(seed=synth.BASE_SEED + 7000 + code * 10 + rows, level=3, rows=rows, nblocks=3,
                               mix=synth.MIX_SINGLE, single_code=code, pwr_min=max(4, min(15, code - 1)),
                               pwr_max=15 if code >= 3 and code <= 16 else 9, val_max=300)
            save(name, d)
            f2.append({"file": name, "code": code, **ref_decode(d)})
    C["F2_codes"] = f2

    # ---- F3: the six invalid codes, and t15/t27/t37 symbols past their range (hand-assembled)
    f3 = []
    for code in synth.BAD_CODES:
        name = "f3_bad%d" % code
        d = synth.generate(seed=1, level=2, rows=3, nblocks=2, mix=synth.MIX_SINGLE, single_code=code)
        save(name, d)
        f3.append({"file": name, **ref_decode(d)})
    for code, width, bad in ((19, 5, 27), (19, 5, 31), (22, 7, 125), (22, 7, 127), (29, 7, 121), (29, 7, 127)):
        # block 0 is fine (all-zero columns), block 1 column 2 carries the out-of-range symbol
        w = BitWriter()
        header(w, 2 * 4 * 4, level=2, rows=4)
        w.put(5, 4), w.put(100, 16)
        for _ in range(4):
            w.put(0, 5)
        w.put(5, 4), w.put(200, 16)
        w.put(0, 5), w.put(0, 5)
        w.put(code, 5), w.put(bad, width)
        w.put(0xFFFFFF, 24)
        name = "f3_code%d_sym%d" % (code, bad)
        d = w.done()
        save(name, d)
        f3.append({"file": name, **ref_decode(d)})
    C["F3_corrupt"] = f3

    # ---- F4: truncation at every byte of a small 2-block file (EOF taxonomy, padding)
    base = synth.generate(seed=synth.BASE_SEED + 4242, level=4, rows=5, nblocks=2, mix=synth.MIX_UNIFORM)
    save("f4_base", base)
    f4 = []
    for n in range(0, len(base) + 1):
        r = ref_decode(base[:n])
        f4.append({"len": n, "open": r["open"], "status": r.get("status"), "words": r.get("words"),
                   "sha256": r.get("sha256"), "raw_tell_end": r.get("raw_tell_end")})
    C["F4_truncation"] = {"file": "f4_base", "cuts": f4}

    # ---- F5: WAVC wrapped copy decodes to the same PCM; broken WAVC prefixes are rejected
    plain = synth.generate(seed=synth.BASE_SEED + 55, level=5, rows=4, nblocks=6, channels=1)
    wavc = synth.generate(seed=synth.BASE_SEED + 55, level=5, rows=4, nblocks=6, channels=1, wavc=1)
    save("f5_plain", plain)
    save("f5_wavc", wavc)
    f5 = {"plain": {"file": "f5_plain", **ref_decode(plain)}, "wavc": {"file": "f5_wavc", **ref_decode(wavc)},
          "wavc_quirk": {"file": "f5_wavc", **ref_decode(wavc, force_chans=-1)},
          "plain_quirk": {"file": "f5_plain", **ref_decode(plain, force_chans=-1)}, "bad": []}
    for label, pos, val in (("tag_D", 3, ord("D")), ("ver_V2", 4, ord("W")), ("ver_1.1", 7, ord("1")), ("hdrlen_29", 16, 29)):
        b = bytearray(wavc)
        b[pos] = val
        name = "f5_wavc_" + label
        save(name, bytes(b))
        f5["bad"].append({"file": name, **ref_decode(bytes(b))})
    C["F5_wavc"] = f5

    # ---- F6: header validation matrix (every rule of decode.c:727-750)
    f6 = []

    def hdr_case(label, **kw):
        w = BitWriter()
        args = dict(total=64, channels=1, rate=22050, level=2, rows=4)
        args.update(kw)
        header(w, **args)
        w.put(5, 4), w.put(7, 16)
        for _ in range(1 << args["level"]):
            w.put(0, 5)
        d = w.done()
        save("f6_" + label, d)
        f6.append({"file": "f6_" + label, **ref_decode(d)})

    hdr_case("ok")
    hdr_case("bad_magic", magic=0x032898)
    hdr_case("bad_version", version=2)
    hdr_case("zero_total", total=0)
    hdr_case("chan0", channels=0)
    hdr_case("chan3", channels=3)
    hdr_case("rate4095", rate=4095)
    hdr_case("rate4096", rate=4096)
    hdr_case("rows0", rows=0)
    hdr_case("level0_rows1", level=0, rows=1, total=5)
    for n in (0, 1, 3, 13):
        d = plain[:n]
        save("f6_short%d" % n, d)
        f6.append({"file": "f6_short%d" % n, **ref_decode(d)})
    C["F6_headers"] = f6

    # ---- F7: API traces
    f7 = {}
    src = synth.generate(seed=synth.BASE_SEED + 77, level=5, rows=6, nblocks=7, channels=2,
                         total_values=7 * 6 * 32 - 7)
    save("f7_src", src)
    # (a) mixed request sizes
    s = O.LibacmStream(O.ref_lib(), src)
    trace = []
    for ask in [1, 2, 3, 5, 64, 7, 8192, 2, 100000, 30, 4, 4096, 4096, 4096, 10, 10]:
        rc, b = s.read(ask)
        trace.append({"ask": ask, "rc": rc, "sha": sha(b), "pcm_tell": s.getter("pcm_tell"),
                      "raw_tell": s.getter("raw_tell"), "time_tell": s.getter("time_tell")})
    s.close()
    f7["reads"] = trace
    # (b) the four output formats + bad wordlen
    fm = []
    for be in (0, 1):
        for sg in (0, 1):
            r = ref_decode(src, be=be, sgned=sg)
            fm.append({"be": be, "sgned": sg, "sha256": r["sha256"], "words": r["words"]})
    s = O.LibacmStream(O.ref_lib(), src)
    f7["formats"] = fm
    f7["bad_wordlen"] = [s.read(64, wordlen=1)[0], s.read(64, wordlen=4)[0]]
    s.close()
    # (c) seeks: forward, backward, to end, past end; PCM that follows; non-seekable source
    s = O.LibacmStream(O.ref_lib(), src)
    sk = []
    for op, arg in [("pcm", 100), ("read", 200), ("pcm", 50), ("read", 64), ("time", 20), ("read", 64), ("pcm", 0),
                    ("read", 32), ("pcm", 660), ("read", 64), ("pcm", 100000), ("read", 64), ("time", 1), ("read", 16)]:
        if op == "read":
            rc, b = s.read(arg)
            sk.append({"op": op, "arg": arg, "rc": rc, "sha": sha(b), "pcm_tell": s.getter("pcm_tell")})
        else:
            rc = s.seek_pcm(arg) if op == "pcm" else s.seek_time(arg)
            sk.append({"op": op, "arg": arg, "rc": rc, "pcm_tell": s.getter("pcm_tell"), "raw_tell": s.getter("raw_tell")})
    s.close()
    f7["seeks"] = sk
    s = O.LibacmStream(O.ref_lib(), src, seekable=False)
    s.read(256)
    f7["noseek_back"] = s.seek_pcm(0)
    f7["noseek_fwd"] = s.seek_pcm(300)
    s.close()
    # (d) getters straight after open (mono, forced channels, no length callback)
    gt = {}
    for label, kw in (("plain", {}), ("force1", {"force_chans": 1}), ("force2", {"force_chans": 2}),
                      ("nolen", {"with_length": False})):
        fc = kw.pop("force_chans", 0)
        s = O.LibacmStream(O.ref_lib(), src, fc, **kw)
        gt[label] = {k: s.getter(k) for k in ("bitrate", "rate", "channels", "raw_total", "raw_tell", "pcm_total",
                                              "pcm_tell", "time_total", "time_tell", "seekable")}
        gt[label]["info"] = s.info()
        s.close()
    f7["getters"] = gt
    # (e) short reads from the callback (the reference treats a short refill near a field as EOF)
    f7["short_reads"] = [{"max_read": m, **{k: v for k, v in ref_decode(src, max_read=m).items()
                                            if k in ("open", "status", "words", "sha256")}}
                         for m in (1, 2, 3, 4, 5, 7, 64, 1000)]
    # (f) read_func failure after N bytes
    f7["read_errors"] = [{"fail_at": n, **{k: v for k, v in ref_decode(src, fail_read_at=n).items()
                                           if k in ("open", "status", "words", "sha256")}}
                         for n in (0, 10, 14)]
    # (g) error strings
    f7["strerror"] = {str(e): O.ref_lib().acm_strerror(e).decode() for e in range(-10, 3)}
    # (h) acm_read_loop swallowing an error after output (util.c:271-273): corrupt second block
    w = BitWriter()
    header(w, 3 * 16, level=2, rows=4)
    w.put(4, 4), w.put(9, 16)
    for c in (3, 0, 18, 0):
        w.put(c, 5)
        if c == 3:
            for _ in range(4):
                w.put(5, 3)
        if c == 18:
            for _ in range(4):
                w.put(0b11, 2)
    w.put(4, 4), w.put(9, 16)
    w.put(0, 5), w.put(1, 5)            # code 1 = corrupt
    d = w.done()
    save("f7_corrupt_block2", d)
    s = O.LibacmStream(O.ref_lib(), d)
    rc1, b1 = s.read(4096, loop=True)
    rc2, _ = s.read(4096, loop=True)
    rc3, _ = s.read(4096, loop=True)
    s.close()
    f7["loop_swallow"] = {"file": "f7_corrupt_block2", "rc": [rc1, rc2, rc3], "sha": sha(b1)}
    C["F7_api"] = f7

    # ---- F8: acmtool transcripts
    f8 = []
    with tempfile.TemporaryDirectory() as td:
        def put(name, data):
            p = os.path.join(td, name)
            with open(p, "wb") as f:
                f.write(data)
            return p

        put("a.acm", plain)
        put("w.acm", wavc)
        put("s.acm", src)
        put("trunc.acm", src[:len(src) // 2])
        put("junk.acm", b"this is not an acm file at all")
        put("bad.acm", open(os.path.join(ACM_DIR, "f3_bad1.acm"), "rb").read())

        def run(label, args, outs=(), keep=False):
            for o in outs:
                if not keep and os.path.exists(os.path.join(td, o)):
                    os.remove(os.path.join(td, o))
            r = subprocess.run([O.REF_TOOL] + args, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            rec = {"label": label, "args": args, "rc": r.returncode,
                   "stdout": base64.b64encode(r.stdout).decode() if len(r.stdout) > 2000 else r.stdout.decode("latin1"),
                   "stdout_b64": len(r.stdout) > 2000, "stdout_sha": sha(r.stdout),
                   "stderr": r.stderr.decode("latin1"), "files": {}}
            for o in outs:
                p = os.path.join(td, o)
                rec["files"][o] = sha(open(p, "rb").read()) if os.path.exists(p) else None
            f8.append(rec)

        run("version", ["-v"])
        run("help", ["-h"])
        run("no_command", ["a.acm"])
        run("two_commands", ["-d", "-i", "a.acm"])
        run("bad_option", ["-Z", "a.acm"])
        run("play_without_libao", ["-p", "a.acm"])
        run("info", ["-i", "a.acm", "w.acm", "s.acm", "junk.acm", "missing.acm"])
        run("info_mono", ["-i", "-m", "s.acm"])
        run("info_stereo", ["-i", "-s", "a.acm"])
        run("info_quiet", ["-i", "-q", "a.acm"])
        run("decode_default", ["-d", "a.acm", "s.acm"], ["a.wav", "s.wav"])
        run("decode_raw", ["-d", "-r", "a.acm"], ["a.raw"])
        run("decode_o", ["-d", "-o", "out.wav", "s.acm"], ["out.wav"])
        run("decode_o_two_files", ["-d", "-o", "out.wav", "a.acm", "s.acm"], ["out.wav"])
        run("decode_stdout", ["-d", "-r", "-o", "-", "a.acm"])
        run("decode_stdout_wav", ["-d", "-o", "-", "a.acm"])
        run("decode_none", ["-d", "-n", "a.acm"], ["a.wav"])
        run("decode_quiet_mono", ["-d", "-q", "-m", "s.acm"], ["s.wav"])
        run("decode_force_stereo", ["-d", "-s", "a.acm"], ["a.wav"])
        run("decode_truncated", ["-d", "trunc.acm"], ["trunc.wav"])
        run("decode_corrupt", ["-d", "bad.acm"], ["bad.wav"])
        run("decode_junk", ["-d", "junk.acm"], ["junk.wav"])
        run("decode_missing", ["-d", "missing.acm"], ["missing.wav"])
        run("decode_nofiles", ["-d"])
        put("m.acm", plain)
        run("set_stereo", ["-S", "m.acm"], ["m.acm"], keep=True)
        run("set_mono", ["-M", "m.acm"], ["m.acm"], keep=True)
        run("set_on_junk", ["-M", "junk.acm", "missing.acm"])
        put("tiny.acm", plain[:5])
        run("set_on_short", ["-S", "tiny.acm"])
    C["F8_cli"] = {"inputs": {"a.acm": "f5_plain", "w.acm": "f5_wavc", "s.acm": "f7_src", "bad.acm": "f3_bad1"},
                   "runs": f8}

    # ---- F9: raw juggle_block vectors (reference's static routine through oracle/ref_probe.c).
    # Inputs are re-derivable (helpers.juggle_inputs, seeded PCG64), so only digests of the
    # reference's outputs are stored.
    from helpers import juggle_inputs
    P = O.refprobe_lib()
    f9 = []
    for level in (1, 2, 3, 7, 9, 10, 11):
        for rows in (1, 3, 17):
            if level >= 9 and rows > 3:
                continue
            cols = 1 << level
            wrap = np.zeros(max(1, 2 * cols - 2), dtype=np.int32)
            ins = juggle_inputs(level, rows)
            rec = {"level": level, "rows": rows, "blocks": [], "pcm": {}}
            last = None
            for blk in ins:
                blk = blk.copy()
                P.refprobe_juggle_block(level, rows, blk.ctypes.data, wrap.ctypes.data)
                rec["blocks"].append({"sha256": sha(blk.tobytes()), "head": [int(x) for x in blk[:8]]})
                last = blk
            for be in (0, 1):
                for sg in (0, 1):
                    dst = np.zeros(rows * cols * 2, dtype=np.uint8)
                    src_blk = last.copy()
                    P.refprobe_output(src_blk.ctypes.data, dst.ctypes.data, rows * cols, level, be, 2, sg)
                    rec["pcm"]["be%d_s%d" % (be, sg)] = sha(dst.tobytes())
            rec["wrap_sha256"] = sha(wrap.tobytes())
            f9.append(rec)
    C["F9_juggle"] = f9

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(G, f, indent=0, sort_keys=True)
    n = len(os.listdir(ACM_DIR))
    size = sum(os.path.getsize(os.path.join(ACM_DIR, x)) for x in os.listdir(ACM_DIR))
    print("wrote %d .acm files (%d bytes), golden.json (%d bytes)" % (
        n, size, os.path.getsize(os.path.join(HERE, "golden.json"))))


if __name__ == "__main__":
    main()
