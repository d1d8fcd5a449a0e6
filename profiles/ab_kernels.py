#!/usr/bin/env python3
"""A/B of experimental builds of libacm_hip.so inside ONE process on one box (GPU box only).

  python3 profiles/ab_kernels.py [--level 9 --rows 16 --blocks 250 --streams 1024] [--rounds 3] [--steps 60]
                                 [--allow-wrong name,...] base.so var1.so var2.so ...

The workload is staged once (the default library's host parser) and uploaded once; every library named on the
command line is dlopen()ed privately (its own device handle, plan and kernels - device pointers are plain HIP
pointers, so the same HBM buffers serve all of them), then the libraries are timed in interleaved rounds with
acmhip_plan_time (HIP events on the launch stream).  After its first launch every library's PCM is compared with
the first library's, word for word (CRC of the whole arena); timing-only builds are named in --allow-wrong.
Prints one line per library: median launch ms, frac of the 8 TB/s roofline (4 B/sample), per-round values.
"""
import argparse
import ctypes as C
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def bind(path):
    from libacm_amd import capi
    L = C.CDLL(os.path.abspath(path))
    vp, sz = C.c_void_p, C.c_size_t
    L.acmhip_last_error.restype = C.c_char_p
    L.acmhip_device_open.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.acmhip_device_sync.argtypes = [vp]
    L.acmhip_download.argtypes = [vp, vp, vp, sz]
    L.acmhip_plan_create.argtypes = [vp, C.POINTER(capi.StreamDesc), sz, C.POINTER(capi.Patch), sz, C.c_uint, C.POINTER(vp)]
    L.acmhip_plan_launch.argtypes = [vp, vp, vp, vp, C.c_uint]
    L.acmhip_plan_time.argtypes = [vp, vp, vp, vp, C.c_uint, C.c_int, C.POINTER(C.c_float)]
    L.acmhip_plan_create_packed.argtypes = [vp, C.POINTER(capi.StreamDesc), sz, C.POINTER(capi.PackedStream), C.POINTER(capi.Patch), sz, C.c_uint, C.POINTER(vp)]
    L.acmhip_plan_bind_mform.argtypes = [vp, vp, vp]
    L.acmhip_mform_tile_rows.argtypes = [C.c_uint]
    L.acmhip_mform_bytes.argtypes = [C.c_uint, C.c_uint64]
    L.acmhip_mform_bytes.restype = C.c_uint64
    L.acmhip_mform_pairs.argtypes = [C.c_uint64]
    L.acmhip_mform_pairs.restype = C.c_uint64
    L.acmhip_mform_rows.argtypes = [C.c_uint, vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]
    return L


class Variant:
    def __init__(self, path, descs, mform=None, d_mform=None, own_form=None):
        from libacm_amd import capi
        path, _, env = path.partition("@")            # lib.so@ACM_K2_ABL=17: environment set around this library's launches
        self.env = dict(kv.split("=", 1) for kv in env.split(",") if kv)
        self.name = os.path.basename(path).replace(".so", "") + ("@" + env if env else "")
        self.L = bind(path)
        if own_form is not None:
            # --own-form: this library's OWN stager writes the byte-plane arena it reads (builds whose forms differ)
            idx, dev0, threads = own_form
            mfa = capi.mform_streams(idx, descs, threads=threads, L=self.L)
            d_mform = mfa.upload(dev0)
            mform = mfa.streams
            self.form_bytes = mfa.nbytes
            print("# %s: its own byte-plane form, %.1f MB, pairs by class code 1 / 2 / 3: %s" % (self.name, mfa.nbytes / 1e6, list(mfa.class_counts()[1:4])), flush=True)
        self.dev = C.c_void_p()
        rc = self.L.acmhip_device_open(0, None, C.byref(self.dev))
        if rc:
            raise SystemExit("%s: device_open %d" % (path, rc))
        arr = (capi.StreamDesc * len(descs))(*descs)
        self.plan = C.c_void_p()
        if mform is not None:
            # --form byteplane: the same byte-plane arena for every library (they must agree on the form's group size)
            # (a library whose byte-plane tiles are smaller than the stager's counts more of them over the same rows)
            pk = (capi.PackedStream * len(descs))(*[capi.PackedStream(m.chunk_off, m.ntiles * 8, m.form) for m in mform])
            rc = self.L.acmhip_plan_create_packed(self.dev, arr, len(descs), pk, None, 0, 0, C.byref(self.plan))
            rc = rc or self.L.acmhip_plan_bind_mform(self.plan, d_mform[0], d_mform[1])
        else:
            rc = self.L.acmhip_plan_create(self.dev, arr, len(descs), None, 0, 0, C.byref(self.plan))
        if rc:
            raise SystemExit("%s: plan_create %d %s" % (path, rc, self.L.acmhip_last_error()))
        self.ms = []

    def setenv(self, on):
        for k, v in self.env.items():
            if on:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)

    def launch(self, bufs):
        rc = self.L.acmhip_plan_launch(self.plan, bufs[0], bufs[1], bufs[2], 0)
        if rc:
            raise SystemExit("%s: launch %d %s" % (self.name, rc, self.L.acmhip_last_error()))

    def sync(self):
        self.L.acmhip_device_sync(self.dev)

    def time(self, bufs, reps):
        ms = C.c_float()
        rc = self.L.acmhip_plan_time(self.plan, bufs[0], bufs[1], bufs[2], 0, reps, C.byref(ms))
        if rc:
            raise SystemExit("%s: time %d" % (self.name, rc))
        return ms.value / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=250)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--allow-wrong", default="")
    ap.add_argument("--form", choices=["int16", "byteplane"], default="int16")
    ap.add_argument("--own-form", action="store_true", help="byte-plane form staged by every library's own stager (builds whose forms differ)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    from libacm_amd import capi, workload
    dev = capi.Device(0)
    b = workload.build_uniform(a.streams, a.level, a.rows, a.blocks, channels=a.channels, threads=max(4, min(64, workload.usable_cpus())))
    bufs = b.upload(dev)
    allow = set(x for x in a.allow_wrong.split(",") if x)
    mf = d_mf = None
    if a.form == "byteplane" and not a.own_form:
        mfa = capi.mform_streams(b.idx, b.descs, threads=max(4, min(64, workload.usable_cpus())))
        d_mf = mfa.upload(dev)
        mf = mfa.streams
    own = (b.idx, dev, max(4, min(64, workload.usable_cpus()))) if a.own_form else None
    vs = [Variant(p, b.descs, mf, d_mf, own) for p in a.libs]
    host = np.empty(b.pcm_words, dtype=np.uint16)
    ref = None
    for v in vs:
        print("# first launch of", v.name, flush=True)
        dev.memset(bufs[2], 0xA5, 2 * b.pcm_words)     # nothing of another library's PCM survives a launch that skips a tile
        dev.sync()
        v.setenv(True)
        v.launch(bufs)
        v.sync()
        v.setenv(False)
        dev.download(host, bufs[2])
        crc = zlib.crc32(host.view(np.uint8))
        if ref is None:
            ref = crc
        v.ok = crc == ref
        if not v.ok and v.name.split("@")[0] not in allow:
            print("# %s: PCM differs from %s" % (v.name, vs[0].name))
    # clock ramp
    for _ in range(40):
        vs[0].launch(bufs)
    vs[0].sync()
    for r in range(a.rounds):
        for v in vs:
            v.setenv(True)
            for _ in range(8):
                v.launch(bufs)
            v.sync()
            v.ms.append(v.time(bufs, a.steps))
            v.setenv(False)
    base = float(np.median(vs[0].ms))
    print("# level %d rows %d blocks %d streams %d: %.1f Msamples per launch" % (a.level, a.rows, a.blocks, a.streams, b.samples / 1e6))
    for v in vs:
        med = float(np.median(v.ms))
        frac = b.samples * 4 / (med * 1e-3) / 8e12
        print("%-28s %s  ms %.4f  frac %.4f  vs base %+.2f%%   [%s]" % (v.name, "ok   " if v.ok else "WRONG", med, frac, (base / med - 1) * 100,
                                                                       " ".join("%.4f" % x for x in v.ms)))


if __name__ == "__main__":
    main()
