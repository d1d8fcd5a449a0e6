#!/usr/bin/env python3
"""Benchmark of the ACM decode hot path on MI355X.

One "step" = one pass of the hot path (amplitude-table unpack -> juggle_block
synthesis -> 16-bit write-out: one acmhip_plan_launch) over one batch of
synthetic ACM streams whose staged form is already resident in HBM.  Default
workload = the configuration BASELINE.json's north_star quotes its target on:
1024 mono streams, acm_level 9, acm_rows 16, 250 blocks each (2.097 Gsamples
per GPU per step; the same sample count as configs[1], which is reported
beside it under `other_levels_kernel_only`).

`--gpus N` without a torchrun environment starts N fresh ranks itself (one
process per GPU, before anything in this process touches a GPU); under
`python -m torch.distributed.run` the ranks come from RANK / WORLD_SIZE.  Every
rank decodes its own 1024 streams (independent streams: no data-path
collective, weak scaling); `value` is the whole-job Msamples/s.

Prints ONE JSON line on rank 0 (contract in the task prompt) including
  roofline     - achieved algorithmic HBM GB/s of the tile kernel
                 (4 B/sample: 2 B staged index in + 2 B PCM out) against 8 TB/s,
                 from HIP events recorded on the launch stream
  cpu_baseline - the same decode on the host cores (the real reference if the
                 prebuilt oracle/_ref is present, else our oracle port), on a
                 bounded sample of the same workload, rank 0 / N=1 only.
Every stream of the workload is compared with the CPU oracle (CRC of its PCM)
before the warm-up and again after the timed region.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 4   # 2 B idx16 read + 2 B PCM16 written
PRECONDITION_S = 0.5        # untimed launches in front of the warm-up (clock ramp), whatever --warmup says
SUSTAINED_STEPS = 300       # the long run reported beside the contract's K steps when K is short
K2_GEOMETRY = {12: (512, 16384), 13: (1024, 32768), 14: (1024, 32768)}      # threads, tile dwords of acm_tile2 (acm_kernels.hip: g_tile2); 256 x 8192 below


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--blocks", type=int, default=250)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (other levels, D2D copy rate, end to end)")
    ap.add_argument("--stagewise", action="store_true", help="force the generic stage-wise kernels")
    ap.add_argument("--form", choices=["auto", "int16", "byteplane", "packed"], default="auto",
                    help="staged form the timed launches read (all three are written by the host stager, include/acm_hip.h): int16 = one "
                         "index per sample; byteplane = every row pair at 4 / 8 / 16 bits per index in matrix-core operand order (levels 7-14: first pass "
                         "on v_mfma_i32_16x16x32_i8); packed = width class per column pair + packed residuals (levels 6-9).  auto = "
                         "byteplane where the level has it, else int16.  The other forms are timed as side measurements")
    ap.add_argument("--packed", action="store_true", help="= --form packed")
    ap.add_argument("--no-packed", action="store_true", help="skip the packed-form side measurement")
    ap.add_argument("--control", choices=["nccl", "gloo"], default="nccl",
                    help="backend of the N > 1 control collectives and of the PCM gather leg: nccl (= RCCL over xGMI, the real thing) or gloo "
                         "(host memory; for rehearsing the N > 1 plumbing on a box without N GPUs)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses GPU 0 (with --control gloo: the rank spawn, CPU pinning, per-rank lines and the gather leg run with "
                         "world > 1 on a one-GPU box; the numbers of such a run mean nothing)")
    ap.add_argument("--workload", choices=["uniform", "corpus"], default="uniform",
                    help="uniform = one shape for every stream (default); corpus = configs[2]: --files mixed "
                         "mono/stereo files, levels 7-9, 1-60 s (sharded by file over the ranks with --gpus N)")
    ap.add_argument("--files", type=int, default=4000)
    ap.add_argument("--no-verify", action="store_true",
                    help="timing experiments with deliberately wrong kernels: skip the oracle check (the line says so)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the PCM gather leg (C2) that is reported beside the headline")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank - exercises the N>1 code path on a 1-GPU box")
    a = ap.parse_args()
    if a.gpus is None:
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))      # torchrun --nproc-per-node N bench.py  ==  --gpus N
    if a.packed:
        a.form = "packed"
    if a.form == "auto":
        # the corpus (levels 7-9) has the byte-plane form throughout
        a.form = "byteplane" if not a.stagewise and (a.workload == "corpus" or 7 <= a.level <= 14) else "int16"
    a.packed = a.form == "packed"
    return a


def visible_gpus():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the GPU


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (this process never touches a GPU)."""
    n = args.gpus
    have = visible_gpus()
    if have < (1 if args.share_device else n):
        raise SystemExit("bench.py: %d GPUs requested, %d visible" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that dies leaves the others waiting in RCCL until its timeout: end them as soon as one fails
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = max(rc, abs(r))
    for p in live:
        p.terminate()
    for p in live:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()
    raise SystemExit(rc)


def pin_rank_cpus(local, local_world):
    """Ranks of one node share the host: give each an equal slice of the usable CPUs (parser / staging threads
    stay next to their GPU's copy engine instead of migrating over the whole box)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        if local_world > 1 and len(cpus) >= local_world:
            per = len(cpus) // local_world
            os.sched_setaffinity(0, cpus[local * per:(local + 1) * per])
    except (AttributeError, OSError):
        pass


def workload_cpus():
    from libacm_amd import workload
    return workload.usable_cpus()


class PowerSampler:
    """Socket power and shader clock of the GPU while the kernel runs, as `rocm-smi --showpower --showclocks` reads them
    (the amdgpu hwmon power1_input file lags by seconds).  The level-9 kernel runs into the package power cap: the clock
    the chip holds under it, not the issue slots or HBM, is what the launch time follows."""

    def __init__(self):
        import shutil
        import threading
        self.tool = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.samples, self._stop, self._th = [], threading.Event(), None

    def _run(self):
        while not self._stop.is_set():
            try:
                r = subprocess.run([self.tool, "--showpower", "--showclocks", "--json"], stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, text=True, timeout=5)
                card = list(json.loads(r.stdout).values())[0]
                w = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
                f = [int(v.strip("()Mhz")) for k, v in card.items() if "sclk clock speed" in k]
                if w and f:
                    # (memory and fabric clocks too: a launch that is slow at a HIGH shader clock and below the cap is waiting for them)
                    other = {k.split()[0]: int(v.strip("()Mhz")) for k, v in card.items()
                             if "clock speed" in k and k.split()[0] in ("mclk", "fclk", "socclk") and v.strip("()Mhz").isdigit()}
                    self.samples.append((w[0], f[0], other))
            except Exception:
                pass

    def start(self):
        import threading
        if self.tool:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join()
        s = [x for x in self.samples if x[1] > 900]         # samples taken while the kernel was running
        if len(s) > 4:
            s = s[1:-1]
        if not s:
            return None
        cap = None
        try:
            r = subprocess.run([self.tool, "--showmaxpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5)
            cap = [float(v) for k, v in list(json.loads(r.stdout).values())[0].items() if "ower" in k][0]
        except Exception:
            pass
        w = sorted(x[0] for x in s)
        f = sorted(x[1] for x in s)
        other = {}
        for k in ("mclk", "fclk", "socclk"):
            v = sorted(x[2][k] for x in s if k in x[2])
            if v:
                other[k + "_mhz_median"] = v[len(v) // 2]
        return {"socket_w_median": round(w[len(w) // 2], 1), "socket_w_max": round(w[-1], 1), "power_cap_w": cap,
                "sclk_mhz_median": f[len(f) // 2], "sclk_mhz_max_of_device": 2400, **other, "samples": len(s),
                "source": "rocm-smi --showpower --showclocks, sampled during a 2 s run of the same launches behind the timed region"}


def copy_ceiling():
    """the box's practical HBM ceiling: best of the 16-byte-per-lane copy kernels of profiles/ubench/copy_bw.hip (read +
    written bytes / time), run in this process through profiles/ubench/libcopybw.so"""
    path = os.path.join(ROOT, "profiles", "ubench", "libcopybw.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.acm_copy_ceiling_gbs.restype = C.c_double
    lib.acm_copy_ceiling_gbs.argtypes = [C.c_size_t, C.c_char_p, C.c_size_t]
    name = C.create_string_buffer(96)
    gbs = lib.acm_copy_ceiling_gbs(4 << 30, name, 96)
    if gbs <= 0:
        return None
    one_way = (C.c_double * 2)()
    lib.acm_one_way_gbs.argtypes = [C.c_size_t, C.POINTER(C.c_double)]
    if lib.acm_one_way_gbs(4 << 30, one_way) != 0:
        one_way = (None, None)
    return round(gbs, 1), name.value.decode(), [round(v, 1) if v else None for v in one_way]


_POISON = None


def poison(d_ptr, nbytes):
    """fill a device buffer with a pattern no decode leaves behind (0xA5 bytes) before another staged form is timed and verified: a
    form whose kernel skipped or mis-addressed tiles must not find the previous form's correct samples in its place"""
    global _POISON
    if _POISON is None:
        lib = C.CDLL(os.path.join(ROOT, "profiles", "ubench", "libcopybw.so"))
        lib.acm_poison.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        _POISON = lib.acm_poison
    if _POISON(d_ptr, nbytes, 0xA5) != 0:
        raise SystemExit("bench.py: could not poison the PCM buffer")


def live_traffic(args, kernel_words=("acm_tile2", "acm_fused", "acm_chunk")):          # "acm_tile2" also matches the packed build, acm_tile2p
    """HBM bytes per launch of the tile kernel, measured on THIS box in THIS run: two child runs of this command under
    rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in passes of their own, nothing else enabled - MI355X_MICROARCH.md "HBM"),
    a handful of launches each with the same staged input.  FETCH_SIZE is doubled (gfx950 tallies its 128-byte requests at
    64 B), both are KiB.  None when rocprofv3 is not there or a pass fails (the line then falls back to the committed profile)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "1", "--no-cpu", "--no-extra", "--no-verify",
             "--streams", str(args.streams), "--level", str(args.level), "--rows", str(args.rows), "--blocks", str(args.blocks),
             "--channels", str(args.channels), "--form", args.form, "--no-packed"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="acm_pmc_", dir="/tmp")
        try:
            # the program itself stands right behind "--": the profiler's library is in the process before it starts
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return None
            total, dispatches = 0.0, set()
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and any(w in row.get("Kernel_Name", "") for w in kernel_words):
                            total += float(row["Counter_Value"])
                            dispatches.add(row["Dispatch_Id"])
            if not dispatches:
                return None
            got[counter] = (total / len(dispatches), len(dispatches))
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return {"hbm_bytes_per_launch": int((2 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024),
            "fetch_size_kib_raw": round(got["FETCH_SIZE"][0], 1), "write_size_kib": round(got["WRITE_SIZE"][0], 1),
            "dispatches": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]], "seconds": round(time.perf_counter() - t0, 1)}


def kernel_source_sha():
    """sha256 of the kernel source with comments and white space stripped (what the compiler sees, line numbers aside): a
    comment edit does not orphan the committed PMC traffic figures, a code edit does"""
    with open(os.path.join(ROOT, "libacm_amd", "csrc", "acm_kernels.hip"), "r", encoding="utf-8", errors="replace") as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"\s+", " ", text)
    return hashlib.sha256(text.encode()).hexdigest()[:16]


def cpu_baseline(batch, budget_s):
    """Decode file images of the workload on the host, one thread, until the budget is used."""
    import oracle_api as O
    files = batch.files
    if not files:
        return None
    kind = "reference" if O.have_ref() else "port"
    words = 0
    t0 = time.perf_counter()
    n = 0
    if kind == "reference":
        lib = O.ref_lib()
        buf = (C.c_uint8 * 16384)()
        for f in files:
            s = O.LibacmStream(lib, f.tobytes())
            while True:                                   # acmtool's decode loop (acmtool.c:274-291)
                rc = lib.acm_read_loop(s.h, buf, 8192, 0, 2, 1)
                if rc <= 0:
                    break
                words += rc // 2
            s.close()
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
    else:
        for f in files:
            w, _ = O.Oracle.decode_discard(f)
            words += w
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    # fill-only share with the port (gives the CPU "synth" stage = total - fill)
    t1 = time.perf_counter()
    fw = 0
    for f in files[:max(1, n // 4)]:
        o = O.Oracle(f.tobytes())
        while o.lib().acmo_fill_next_block(o.h, None, None, None) == 1:
            fw += o.getter("block_len")
        o.close()
    dt_fill = time.perf_counter() - t1
    # the same decoder on all usable host cores, one stream per thread; big reads keep the GIL out of it
    all_cores = None
    if kind == "reference":
        from concurrent.futures import ThreadPoolExecutor
        ncpu = workload_cpus()
        stop_at = time.perf_counter() + min(6.0, budget_s / 2)

        def one(f):
            if time.perf_counter() > stop_at:
                return 0
            s = O.LibacmStream(lib, f.tobytes())
            big = (C.c_uint8 * (1 << 20))()
            w = 0
            while True:
                rc = lib.acm_read_loop(s.h, big, 1 << 20, 0, 2, 1)
                if rc <= 0:
                    break
                w += rc // 2
            s.close()
            return w
        t2 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncpu) as ex:
            aw = sum(ex.map(one, files))
        dt_all = time.perf_counter() - t2
        all_cores = {"value": round(aw / dt_all / 1e6, 1), "unit": "Msamples/s", "cores": ncpu,
                     "sample": "%.1f Msamples in %.1f s, one stream per thread" % (aw / 1e6, dt_all)}
    return {
        "all_cores": all_cores,
        "value": round(words / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "sample": "%d of the workload's streams (%.1f Msamples), whole decode path incl. bit parsing, 1 thread, %.1f s"
                  % (n, words / 1e6, dt),
        "fill_only_msamples_s": round(fw / dt_fill / 1e6, 2) if dt_fill > 0 else None,
        "host_cpus": os.cpu_count(), "usable_cpus": workload_cpus(),
    }


def oracle_crcs(batch, threads):
    """CRC-32 of every stream's PCM as the CPU oracle decodes it (the checker; untimed setup)."""
    import oracle_api as O
    from concurrent.futures import ThreadPoolExecutor

    def one(k):
        pcm = O.Oracle.decode_all(batch.files[k].tobytes())[0]
        return zlib.crc32(pcm.view(np.uint8)[:2 * batch.descs[k].n_emit])
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(one, range(len(batch.files))))


def device_crcs(dev, batch, d_pcm, threads):
    """CRC-32 of every stream's PCM as it sits in HBM (downloaded in slices of whole streams)."""
    from concurrent.futures import ThreadPoolExecutor
    out = [0] * len(batch.descs)
    order = sorted(range(len(batch.descs)), key=lambda k: batch.descs[k].pcm_off)
    slab_words = 1 << 28
    with ThreadPoolExecutor(max_workers=threads) as ex:
        i = 0
        while i < len(order):
            lo = batch.descs[order[i]].pcm_off
            j = i
            while j < len(order) and batch.descs[order[j]].pcm_off + batch.descs[order[j]].n_emit - lo <= slab_words:
                j += 1
            j = max(j, i + 1)
            hi = batch.descs[order[j - 1]].pcm_off + batch.descs[order[j - 1]].n_emit
            host = np.empty(hi - lo, dtype=np.uint16)
            dev.download(host, d_pcm + 2 * lo)
            raw = host.view(np.uint8)

            def one(k):
                d = batch.descs[k]
                return k, zlib.crc32(raw[2 * (d.pcm_off - lo): 2 * (d.pcm_off - lo + d.n_emit)])
            for k, c in ex.map(one, order[i:j]):
                out[k] = c
            i = j
    return out


def time_plan(dev, plan, bufs, steps, warmup, barrier):
    """W warm-up + K timed launches; returns (wall seconds incl. sync brackets, device ms from HIP events)."""
    d_idx, d_hdr, d_pcm = bufs
    for _ in range(warmup):
        plan.launch(d_idx, d_hdr, d_pcm)
    dev.sync()
    barrier()
    t0 = time.perf_counter()
    ev_ms = plan.time(d_idx, d_hdr, d_pcm, reps=steps)    # events on the launch stream; blocks until done
    dev.sync()
    barrier()
    return time.perf_counter() - t0, ev_ms


def precondition(dev, plan, bufs, seconds):
    """untimed launches until `seconds` have passed (the chip reaches the clock it holds under this load)"""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            plan.launch(*bufs)
        dev.sync()
        n += 10
    return n


def side_measure(dev, capi, workload, level, rows, blocks, streams, steps, verify=256, channels=1, packed=True, corpus_files=0):
    """kernel-only rate of another configuration (configs[1] is level 7; the stress config is level 11) on every staged form its
    level has: the byte-plane form where there is one (that is the entry's own rate), the int16 form, the packed form.  The PCM
    each set of timed launches leaves behind is compared with the CPU oracle on the first `verify` streams (CRC-32 each)."""
    import oracle_api as O
    from concurrent.futures import ThreadPoolExecutor
    if corpus_files:
        # configs[2]: the 4000-file corpus (levels 7-9, mono / stereo, ragged) in ONE plan; the first `verify` files are checked
        b = workload.build_corpus(corpus_files, keep_files=verify, threads=workload.usable_cpus())
        level, rows, blocks, streams, channels, packed = 9, 16, 0, corpus_files, 0, False
    else:
        b = workload.build_uniform(streams, level, rows, blocks, channels=channels, seed0=1 << 20, keep_files=verify)
    bufs = b.upload(dev)
    mf = d_mf = None
    if capi.lib().acmhip_mform_tile_rows(level) > 0:
        mf = capi.mform_streams(b.idx, b.descs, threads=workload.usable_cpus())
        d_mf = mf.upload(dev)
        bps = round(mf.nbytes / b.samples, 3)
        mf.data = None                  # the host copy has done its job (configs[4]: 34 GB)
    plan = capi.Plan(dev, b.descs, packed=mf.streams if mf else None)
    tiles = plan.stats().tiles

    def check():
        if not b.files:
            return 0
        d_last = b.descs[len(b.files) - 1]
        host = np.empty(d_last.pcm_off + d_last.n_emit, dtype=np.uint16)
        dev.download(host, bufs[2])
        raw = host.view(np.uint8)

        def one(k):
            d = b.descs[k]
            want = O.Oracle.decode_all(b.files[k].tobytes())[0]
            return zlib.crc32(want.view(np.uint8)[:2 * d.n_emit]) == zlib.crc32(raw[2 * d.pcm_off: 2 * (d.pcm_off + d.n_emit)])
        with ThreadPoolExecutor(max_workers=max(4, min(64, workload.usable_cpus()))) as ex:
            ok = list(ex.map(one, range(len(b.files))))
        if not all(ok):
            raise RuntimeError("side measurement level %d: HIP output differs from the oracle on %d of %d streams" % (level, ok.count(False), len(ok)))
        return len(ok)

    def timed(pl):
        poison(bufs[2], 2 * b.pcm_words)
        precondition(dev, pl, bufs, 0.2)
        _, ms = time_plan(dev, pl, bufs, steps, 5, lambda: None)
        rate = b.samples * steps / (ms * 1e-3)
        return {"msamples_s": round(rate / 1e6, 1), "algo_gbs": round(rate * ALGO_BYTES_PER_SAMPLE / 1e9, 1),
                "frac_hbm": round(rate * ALGO_BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS, 4), "launch_ms": round(ms / steps, 4),
                "verified_streams": check()}
    out = {"level": level, "rows": rows, "streams": streams, "blocks": blocks, "channels": channels, "steps": steps, "tiles": int(tiles)}
    if corpus_files:
        out.update(level="7-9", blocks="ragged", channels="1|2", msamples_per_launch=round(b.samples / 1e6, 1),
                   config="BASELINE.json configs[2]: %d-file corpus in one plan" % corpus_files)
        form = sum(min(plan.form_rows(i) << d.level, d.n_emit) for i, d in enumerate(b.descs))
        out["samples_from_byteplane_form"] = round(form / max(1, int(b.samples)), 5)
    int16 = timed(plan)
    if mf is not None:
        plan.bind_mform(*d_mf)
        out.update(timed(plan), staged_form="byteplane", staged_bytes_per_sample=bps)
        out["int16_form"] = int16
        for p_ in d_mf:
            dev.free(p_)
    else:
        out.update(int16, staged_form="int16")
    plan.destroy()
    if packed and capi.packed_tile_rows(level) > 0:
        pk = capi.pack_streams(b.idx, b.descs, threads=workload.usable_cpus())
        pk_ptrs = pk.upload(dev)
        plan = capi.Plan(dev, b.descs, packed=pk.streams)
        plan.bind_packed(*pk_ptrs)
        t = timed(plan)
        out["packed_form"] = {"frac_hbm": t["frac_hbm"], "verified_streams": t["verified_streams"], "staged_bytes_per_sample": round(pk.nbytes / b.samples, 3)}
        plan.destroy()
        for p in pk_ptrs:
            dev.free(p)
    for p in bufs:
        dev.free(p)
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                  # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    have = visible_gpus()
    if have < (1 if args.share_device else local_world):
        raise SystemExit("bench.py: %d GPUs requested, %d visible" % (local_world, have))
    pin_rank_cpus(local, local_world)
    if args.share_device:
        local = 0                                          # every rank on GPU 0 (a rehearsal of the N > 1 plumbing, not a measurement)
    if args.control == "gloo" and not args.share_device and world > 1:
        raise SystemExit("bench.py: --control gloo is for rehearsals (--share-device); real N > 1 runs use RCCL")

    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "RANK" not in os.environ:                       # --force-dist outside a launcher: a world of one
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]))
            s.close()
        torch.cuda.set_device(local)
        if args.control == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from libacm_amd import capi, workload
    if capi.device_count() <= 0:
        raise SystemExit("bench.py: no HIP device - the hot path has no CPU fallback")
    dev = capi.Device(local)

    cdev = "cpu" if args.control == "gloo" else "cuda:%d" % local          # where the control tensors (and, with gloo, the gathered PCM) live
    sync_t = torch.zeros(1, device=cdev) if dist is not None else None

    def barrier():
        if dist is not None:
            dist.all_reduce(sync_t)          # RCCL barrier (control only: no data-path collective)
            torch.cuda.synchronize()

    # ---- stage the workload (untimed): synth -> host bit parsing -> HBM ----
    # file images are kept for the oracle check of EVERY stream (and, on rank 0 at N=1, for the CPU baseline)
    t0 = time.perf_counter()
    stage_threads = max(4, min(64, workload_cpus()))       # the rank's own slice of the host (pin_rank_cpus)
    keep = 0 if args.no_verify and (args.no_cpu or world != 1) else 1 << 30
    if args.workload == "corpus":
        # configs[2]/[3]: the SAME corpus whatever N; rank r decodes its longest-first shard of the file list
        from libacm_amd import batch as fe
        shapes = workload.corpus_shapes(args.files)
        mine = fe.shard_longest_first([s["total_values"] for s in shapes], world)[rank]
        batch = workload.build_corpus(len(mine), shapes=[shapes[i] for i in mine], seed0=0, keep_files=keep,
                                      threads=stage_threads)
    else:
        batch = workload.build_uniform(args.streams, args.level, args.rows, args.blocks, channels=args.channels,
                                       seed0=rank * args.streams, keep_files=keep, threads=stage_threads)
    t_stage = time.perf_counter() - t0
    if dist is not None:
        # the PCM of a multi-rank run lives in a torch tensor so that the gather leg can hand it to RCCL
        d_idx = dev.malloc(batch.idx.nbytes)
        d_hdr = dev.malloc(batch.hdr.nbytes)
        pcm_t = torch.empty(batch.pcm_words, dtype=torch.int16, device="cuda:%d" % local)
        torch.cuda.synchronize()
        flat = batch.idx.view(np.uint8)
        for o in range(0, flat.size, 1 << 28):
            dev.upload(d_idx + o, flat[o:o + (1 << 28)])
        dev.upload(d_hdr, batch.hdr)
        bufs = (d_idx, d_hdr, pcm_t.data_ptr())
    else:
        pcm_t = None
        bufs = batch.upload(dev)
    # the second staged form of the same streams (host stager, untimed like the parsing), bound for the headline launches:
    # --form byteplane (the default where the level has it) or packed; the other forms are timed as side measurements behind them
    pk = pk_ptrs = mf = d_mf = None
    t_pack = t_mform = None
    if args.form == "packed":
        if args.stagewise or args.workload != "uniform" or capi.packed_tile_rows(args.level) <= 0:
            raise SystemExit("bench.py: --form packed needs a uniform workload of a level with a packed form (6-9)")
        t0 = time.perf_counter()
        pk = capi.pack_streams(batch.idx, batch.descs, threads=stage_threads)
        t_pack = time.perf_counter() - t0
        pk_ptrs = pk.upload(dev)
    elif args.form == "byteplane":
        if args.stagewise or (args.workload == "uniform" and capi.lib().acmhip_mform_tile_rows(args.level) <= 0):
            raise SystemExit("bench.py: --form byteplane needs a level with a byte-plane form (7-14)")
        t0 = time.perf_counter()
        mf = capi.mform_streams(batch.idx, batch.descs, threads=stage_threads)
        t_mform = time.perf_counter() - t0
        d_mf = mf.upload(dev)
    second = pk.streams if pk else mf.streams if mf else None
    plan = capi.Plan(dev, batch.descs, flags=capi.PLAN_STAGEWISE if args.stagewise else capi.PLAN_AUTO, packed=second)

    def bind_headline_form():
        if pk:
            plan.bind_packed(*pk_ptrs)
        elif mf:
            plan.bind_mform(*d_mf)
    bind_headline_form()
    stats = plan.stats()

    # setup-time check (untimed): every stream of the workload against the CPU oracle, CRC-32 of its PCM
    verified, n_verified, want = None, 0, None
    if batch.files and not args.no_verify:
        want = oracle_crcs(batch, stage_threads)
        plan.launch(*bufs)
        dev.sync()
        got = device_crcs(dev, batch, bufs[2], stage_threads)
        bad = [k for k in range(len(want)) if want[k] != got[k]]
        if bad:
            raise SystemExit("bench.py: HIP output differs from the oracle on %d of %d streams (first: %d) - "
                             "refusing to report a number" % (len(bad), len(want), bad[0]))
        verified, n_verified = True, len(want)

    pre = precondition(dev, plan, bufs, PRECONDITION_S)
    wall, ev_ms = time_plan(dev, plan, bufs, args.steps, args.warmup, barrier)
    sustained = None
    if args.steps < SUSTAINED_STEPS:
        # the contract's K is short (tens of ms): the same loop again, long, right behind it
        swall, sev = time_plan(dev, plan, bufs, SUSTAINED_STEPS, 0, barrier)
        sustained = {"steps": SUSTAINED_STEPS, "ms_per_step": round(swall / SUSTAINED_STEPS * 1e3, 4),
                     "launch_ms": round(sev / SUSTAINED_STEPS, 4),
                     "msamples_s_per_gpu": round(batch.samples * SUSTAINED_STEPS / swall / 1e6, 1),
                     "frac": round(batch.samples * ALGO_BYTES_PER_SAMPLE / (sev / SUSTAINED_STEPS * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # the same streams on the other staged forms, right behind (every stream's PCM compared with the oracle's again)
    other_forms = []
    if not args.no_extra and not args.stagewise:
        nsteps = max(args.steps, min(SUSTAINED_STEPS, 100))

        def time_form(name, pl, extra):
            poison(bufs[2], 2 * batch.pcm_words)        # nothing of the previous form's (correct) PCM may survive a launch that skips tiles
            _, oev = time_plan(dev, pl, bufs, nsteps, 5, lambda: None)
            oms = oev / nsteps
            o = {"form": name, "launch_ms": round(oms, 4), "steps": nsteps,
                 "frac": round(batch.samples * ALGO_BYTES_PER_SAMPLE / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            o.update(extra)
            if want is not None:
                got = device_crcs(dev, batch, bufs[2], stage_threads)
                o["verified_streams"] = sum(1 for a_, b_ in zip(got, want) if a_ == b_)
                if got != want:
                    raise SystemExit("bench.py: PCM of the %s form differs from the oracle - refusing to report a number" % name)
            other_forms.append(o)
        if args.form != "int16":
            # the headline's own plan with nothing bound reads the int16 arena (same tiles, acm_tile2's vector-ALU build)
            plan.bind_packed(None, None)
            plan.bind_mform(None, None)
            time_form("int16 index per sample", plan, {})
            bind_headline_form()
        if args.form != "byteplane" and args.workload == "uniform" and capi.lib().acmhip_mform_tile_rows(args.level) > 0:
            t0 = time.perf_counter()
            mf2 = capi.mform_streams(batch.idx, batch.descs, threads=stage_threads)
            tm_ = time.perf_counter() - t0
            d2 = mf2.upload(dev)
            p2 = capi.Plan(dev, batch.descs, packed=mf2.streams)
            p2.bind_mform(*d2)
            time_form("byteplane (row pairs at 8 / 12 / 16 bits per index - 4 / 8 / 16 at level 7 - in matrix-core operand order)", p2,
                      {"host_reorder_seconds": round(tm_, 2), "staged_bytes_per_sample": round(mf2.nbytes / batch.samples, 3)})
            p2.destroy()
            for p_ in d2:
                dev.free(p_)
            del mf2
        if args.form != "packed" and not args.no_packed and args.workload == "uniform" and capi.packed_tile_rows(args.level) > 0 and world == 1:
            t0 = time.perf_counter()
            pk2 = capi.pack_streams(batch.idx, batch.descs, threads=stage_threads)
            tp_ = time.perf_counter() - t0
            ptr2 = pk2.upload(dev)
            p2 = capi.Plan(dev, batch.descs, packed=pk2.streams)
            p2.bind_packed(*ptr2)
            time_form("packed (class per column pair + packed residuals)", p2,
                      {"staged_bytes_per_sample_packed": round(pk2.nbytes / batch.samples, 3), "host_pack_seconds": round(tp_, 2)})
            p2.destroy()
            for p_ in ptr2:
                dev.free(p_)
            del pk2
        plan.launch(*bufs)          # the headline form's PCM is what the final check below looks at
        dev.sync()

    # what the chip's power and clock read while this kernel runs (untimed: 2 s of the same launches)
    power = None
    if rank == 0 and world == 1 and not args.no_extra:
        sampler = PowerSampler()
        sampler.start()
        precondition(dev, plan, bufs, 2.0)
        power = sampler.stop()

    # the timed launches must have left exactly the verified PCM behind (the hot path carries no state between launches)
    if want is not None:
        got = device_crcs(dev, batch, bufs[2], stage_threads)
        if got != want:
            raise SystemExit("bench.py: PCM after the timed region differs from the oracle - refusing to report a number")

    # the box's copy kernel between THESE arenas (staged input -> PCM arena; garbage PCM, behind the last check): the rate of a launch
    # follows where its arenas landed in physical memory by +-5 % (profiles/r5_placement.txt: a plain copy between the same two
    # allocations moves with it), so the like-for-like ceiling is a copy between the very buffers that were timed
    copy_same = None
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            cb = C.CDLL(os.path.join(ROOT, "profiles", "ubench", "libcopybw.so"))
            cb.acm_copy_between_gbs.restype = C.c_double
            cb.acm_copy_between_gbs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
            src, src_bytes = (d_mf[0], mf.data.nbytes) if mf else (bufs[0], 2 * batch.pcm_words)
            nb = min(src_bytes, 2 * batch.pcm_words) // (1 << 20) * (1 << 20)
            dev.sync()
            copy_same = round(cb.acm_copy_between_gbs(bufs[2], src, nb), 1) if nb else None
        except Exception:
            copy_same = None

    # max over ranks of the bracketed wall time
    if dist is not None:
        t = torch.tensor([wall], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t.item())
        s = torch.tensor([float(batch.samples), float(n_verified)], device=cdev, dtype=torch.float64)
        dist.all_reduce(s)
        total_samples, n_verified = float(s[0].item()), int(s[1].item())
        # why the N-rank number is what it is: every rank's own launch time and sample count (the corpus is sharded
        # longest-first by header weight; a rank that got more samples, or a slower chip, sets the job's time)
        mine = torch.tensor([ev_ms / args.steps, float(batch.samples), float(len(batch.descs))], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "launch_ms": round(float(e[0]), 4), "msamples": round(float(e[1]) / 1e6, 1), "streams": int(e[2])}
                    for r, e in enumerate(every)]
    else:
        per_rank = None
        wall_max, total_samples = wall, float(batch.samples)

    # ---- C2 (N > 1): gather of every rank's PCM into rank 0's HBM over RCCL/xGMI, reported beside the headline
    gather = None
    if dist is not None and world > 1 and not args.no_gather:
        sizes = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([batch.pcm_words], dtype=torch.int64, device=cdev))
        sizes = [int(x.item()) for x in sizes]
        wire = pcm_t.view(torch.uint8)
        if args.control == "gloo":
            wire = wire.cpu()                              # (a rehearsal: gloo moves host memory)
        recv = [torch.empty(2 * sizes[r], dtype=torch.uint8, device=cdev) for r in range(1, world)] if rank == 0 else []
        gt = []
        for rep in range(3):
            barrier()
            g0 = time.perf_counter()
            if rank == 0:
                reqs = [dist.irecv(recv[r - 1], src=r) for r in range(1, world)]
            else:
                reqs = [dist.isend(wire, dst=0)]
            for q in reqs:
                q.wait()
            torch.cuda.synchronize()
            barrier()
            gt.append(time.perf_counter() - g0)
        g = min(gt[1:])
        moved = 2 * sum(sizes[1:])
        gather = {"seconds": round(g, 4), "bytes_into_rank0": moved, "gbs": round(moved / g / 1e9, 1),
                  "msamples_s_with_gather": round(total_samples / (wall_max / args.steps + g) / 1e6, 1)}
        del recv

    value = total_samples * args.steps / wall_max / 1e6
    ms_per_step = wall_max / args.steps * 1e3
    launch_ms = ev_ms / args.steps                         # average duration of one launch (rank 0's own events)
    achieved = batch.samples * ALGO_BYTES_PER_SAMPLE / (launch_ms * 1e-3) / 1e9

    # HBM traffic per launch from the committed PMC run of this same command (profiles/run_profile.sh): bench.py cannot
    # collect counters on itself.  The file records the kernel source it was measured on; another source -> null.
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r6_traffic.json")) as f:
            tj = json.load(f)
        key = "level%d_%dx%dblocks_rows%d" % (args.level, args.streams, args.blocks, args.rows)
        if args.channels != 1:
            key += "_ch%d" % args.channels
        if args.form != "int16":
            key += "_" + args.form
        if args.workload == "uniform" and tj.get("kernel_source_sha16") == kernel_source_sha() and key in tj:
            traffic = tj[key]["hbm_bytes_per_launch"]
            traffic_src = "profiles/r6_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE, same kernel source %s)" % tj["kernel_source_sha16"]
    except Exception:
        traffic = None

    # ... and, on rank 0 of a one-GPU default run, measured now: two short child runs of this command under rocprofv3 --pmc
    traffic_committed = traffic
    if rank == 0 and world == 1 and not args.no_extra and not args.stagewise and args.workload == "uniform" and 6 <= args.level <= 14:
        lt = live_traffic(args)
        if lt:
            traffic = lt["hbm_bytes_per_launch"]
            traffic_src = ("measured in this run on this box: two child runs of this command under rocprofv3 --pmc (FETCH_SIZE x2, WRITE_SIZE; "
                           "%d / %d dispatches of the tile kernel, %.0f s)" % (lt["dispatches"][0], lt["dispatches"][1], lt["seconds"]))

    # the same launch priced on the bytes it really moved (the roofline fraction above stays on the algorithmic 4 B/sample)
    frac_measured = round(traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None

    lv_txt = "7-9" if args.workload == "corpus" else args.level
    shape = (args.streams, args.level, args.rows, args.blocks, args.channels)
    out = {
        "metric": "decoded PCM Msamples/sec over a batch of ACM streams (hot path on HBM-resident staged input)",
        "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if args.workload == "corpus" else "weak", "vs_baseline": None,
        "verified_vs_oracle": verified if not args.no_verify else "skipped (--no-verify)",
        "verified_streams": n_verified, "dtype": "int32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[2]: %d-file corpus, mixed mono/stereo, acm_level 7-9, 1-60 s at 22050 Hz, "
                                "sharded by file" % args.files) if args.workload == "corpus" else
                               "%d synthetic %s streams per GPU, acm_level %d, acm_rows %d, %d blocks each%s" % (
                                   args.streams, "mono" if args.channels == 1 else "stereo", args.level, args.rows, args.blocks,
                                   " (north_star target configuration)" if shape == (1024, 9, 16, 250, 1) else
                                   " (BASELINE.json configs[1])" if shape == (1024, 7, 16, 1000, 1) else ""),
                   "streams_per_gpu": len(batch.descs), "acm_level": lv_txt,
                   "acm_rows": args.rows, "blocks_per_stream": "ragged" if args.workload == "corpus" else args.blocks,
                   "channels": "1|2" if args.workload == "corpus" else args.channels,
                   "samples_per_step_per_gpu": int(batch.samples), "sharding": "streams (independent), no collective",
                   "kernel": "stagewise" if args.stagewise else "acm_tile2 + acm_fused_tile", "tiles": int(stats.tiles),
                   "launches_per_step": int(stats.launches), "host_stage_seconds": round(t_stage, 2),
                   "staged_form": ("packed: width class per column pair and 16-row group + residuals at 0/4/8/16 bits (%.3f B/sample), "
                                   "{val, pwr} per block; written by the host stager (acmhip_pack_tiles)" % (pk.nbytes / batch.samples))
                                  if pk else
                                  ("byteplane: every row pair's indices at the narrowest width class that holds them - 8 / 12 / 16 bits at a level of the chunk "
                                   "kernel (12: a signed low byte + a signed high nibble; 16: two signed bytes), 4 / 8 / 16 at level 7; %.3f B/sample here; "
                                   "pairs by class code 1 (12 bits; 4 at level 7) / 2 (8 bits) / 3 (16 bits): %d / %d / %d), per row in groups of %s columns a residue class apart (the order the "
                                   "matrix cores read operands in) + a 4-byte entry per pair + {val, pwr} per block; written by the host stager "
                                   "(acm_stage_file + acmhip_mform_rows, %.2f s for this batch)" % (
                                       (mf.nbytes / batch.samples,) + tuple(int(x) for x in mf.class_counts()[1:4]) +
                                       ("8, 16 or 64" if args.workload == "corpus" else str(capi.lib().acmhip_mform_group(args.level)), t_mform))) if mf else
                                  "int16 index per sample + {val, pwr} per block, written by the host stager (acm_stage_file)",
                   "untimed_precondition_launches": pre},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_committed_profile": traffic_committed, "frac_of_peak_on_measured_traffic": frac_measured,
                     "algorithmic_bytes_per_launch": int(batch.samples) * ALGO_BYTES_PER_SAMPLE,
                     "kernel": (("acm_chunk<%s>: six stages on v_mfma_i32_16x16x64_i8, one wavefront per 2048-sample chunk, one LDS pass group behind; "
                                 "chunks inside one block on the scalar fast path (+ acm_fused_tile on ragged tails)" % lv_txt)
                                if mf and args.workload == "uniform" and capi.lib().acmhip_mform_group(args.level) == 64 and args.level <= 12 else
                                ("acm_tile2<TileCfg<%s,%d,%d>, first six stages on v_mfma_i32_16x16x64_i8 (a row pair per tile, sixteen residue classes per "
                                 "wavefront), LDS passes with barriers behind> (+ the prefix + plane pair on ragged tails)" % (
                                     (lv_txt,) + {13: (512, 16384), 14: (1024, 32768)}[args.level]))
                                if mf and args.workload == "uniform" and capi.lib().acmhip_mform_group(args.level) == 64 else
                                "acm_tile2%s<TileCfg<%s,%d,%d>%s> (+ acm_fused_tile on ragged tails)" % (
                                    ("p" if pk else "", lv_txt) + K2_GEOMETRY.get(args.level, (256, 8192)) +
                                    (", first pass on the matrix cores (acm_chunk at the levels that have it)" if mf and args.workload == "corpus" else
                                     ", first pass on v_mfma_i32_16x16x32_i8" if mf else "",))
                                if args.workload == "corpus" or 6 <= args.level <= 14 else "see DESIGN.md section 2 for level %s" % lv_txt),
                     "launch_ms": round(launch_ms, 4), "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                     # the same launch on the bytes it really moved (task 6): `achieved` / `frac` price SURVEY's canonical 4 B/sample
                     "achieved_real_gbs": round(traffic / (launch_ms * 1e-3) / 1e9, 1) if traffic else None},
    }
    # which kernel family decodes what share of the samples (task 3): rows a plan reads from the byte-plane form go to
    # acm_chunk (levels 8-12) / acm_tile2's matrix builds (7, 13, 14); everything else - ragged tails, streams the form could not take -
    # is read as int16 by acm_fused_tile and friends.  Asked of the plan stream by stream (acmhip_plan_form_rows), not assumed.
    if mf and rank == 0:
        by_level, form_total = {}, 0
        for i, d in enumerate(batch.descs):
            fr = plan.form_rows(i) << d.level
            fr = min(fr, d.n_emit)
            e = by_level.setdefault(int(d.level), [0, 0])
            e[0] += fr
            e[1] += d.n_emit
            form_total += fr
        out["config"]["kernel_share"] = {
            "samples_from_byteplane_form": round(form_total / max(1, int(batch.samples)), 5),
            "samples_from_int16_form": round(1.0 - form_total / max(1, int(batch.samples)), 5),
            "by_level": {str(lv): round(a / max(1, b), 5) for lv, (a, b) in sorted(by_level.items())},
            "note": "byte-plane rows: acm_chunk (levels 8-12) or acm_tile2's matrix build (7, 13, 14); int16 rows: ragged tails and streams "
                    "without a form (H1 patches, levels below 7 and level 15) on acm_fused_tile / the register and stage-wise kernels"}
    if args.share_device or args.control == "gloo":
        out["rehearsal"] = ("every rank on GPU 0, control and gather through gloo: the N > 1 plumbing (rank spawn, CPU slices, per-rank lines, "
                            "gather leg) is exercised, the numbers mean nothing")
    if world == 1:
        out["multi_gpu"] = ("no N > 1 run on hardware exists for this code: the pool it is developed on hands out one GPU at a time "
                            "(N > 1 is covered by gloo tests on CPU with up to eight ranks and by a two-rank rehearsal on one GPU)")
    if per_rank:
        ms = [p["launch_ms"] for p in per_rank]
        sm = [p["msamples"] for p in per_rank]
        out["per_rank"] = per_rank
        out["imbalance"] = {"launch_ms_max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4),
                            "samples_max_over_mean": round(max(sm) / (sum(sm) / len(sm)), 4)}
    if other_forms:
        out["other_staged_forms"] = other_forms
        if mf:
            # north_star says "no MFMA"; VERDICT r3 (task 8) allowed the matrix-core first pass as an experiment to keep if it wins.  The
            # headline is that build; the same streams on the int16 form run the vector-ALU build of the same kernel - its figure, measured
            # in this run, sits right beside the headline's for whoever wants to price the path without the matrix cores
            i16 = [o for o in other_forms if o["form"].startswith("int16")]
            if i16:
                out["roofline"]["frac_vector_alu_build_int16_form"] = i16[0]["frac"]
                out["roofline"]["launch_ms_vector_alu_build_int16_form"] = i16[0]["launch_ms"]
    if sustained:
        out["sustained"] = sustained
    if power:
        out["power"] = power
    if gather:
        out["gather_c2"] = gather

    if rank == 0 and world == 1:
        if not args.no_extra:
            # the box's own device-to-device copy rate (read + write bytes), the practical HBM ceiling
            try:
                x = torch.empty(1 << 29, dtype=torch.int32, device="cuda")
                y = torch.empty_like(x)
                y.copy_(x)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(5):
                    y.copy_(x)
                e1.record()
                torch.cuda.synchronize()
                copy_gbs = 5 * 2 * x.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                out["roofline"]["torch_copy_gbs"] = round(copy_gbs, 1)
                del x, y
                torch.cuda.empty_cache()
            except Exception:
                out["roofline"]["torch_copy_gbs"] = None
            try:
                best = copy_ceiling()
                if best:
                    out["roofline"]["d2d_copy_gbs"] = best[0]
                    out["roofline"]["d2d_copy_kernel"] = best[1] + " (profiles/ubench/copy_bw.hip, best of eight 16-B-per-lane copy kernels)"
                    out["roofline"]["frac_of_d2d_copy"] = round(achieved / best[0], 4)
                    # ... and the honest one: measured bytes against the copy kernel's measured bytes
                    out["roofline"]["frac_of_d2d_copy_on_measured_traffic"] = (
                        round(traffic / (launch_ms * 1e-3) / 1e9 / best[0], 4) if traffic else None)
                    out["roofline"]["read_only_gbs"], out["roofline"]["write_only_gbs"] = best[2]
                if copy_same:
                    out["roofline"]["d2d_copy_same_arenas_gbs"] = copy_same
                    out["roofline"]["frac_of_d2d_copy_same_arenas_on_measured_traffic"] = (
                        round(traffic / (launch_ms * 1e-3) / 1e9 / copy_same, 4) if traffic else None)
            except Exception as e:
                out["roofline"]["d2d_copy_gbs"] = None
                out["roofline"]["d2d_copy_error"] = str(e)[:120]
        if not args.no_extra and not args.stagewise:
            extra = []
            # configs[1] and a level-11 batch with the headline's sample count; configs[4] at its full size (65 536 distinct stereo
            # streams, 17.2 Gsamples, 34 GB of staged indices + 34 GB of PCM in HBM; ~1 min of staging)
            # ... and levels 10, 12, 13, 14 with the headline's sample count (task 2: a driver-side number for each)
            for (lv, rw, bl, ns, ch) in ((7, 16, 1000, 1024, 1), (11, 64, 16, 1024, 1), (10, 16, 125, 1024, 1), (12, 64, 8, 1024, 1), (13, 64, 4, 1024, 1),
                                         (14, 8, 16, 1024, 1), (11, 64, 2, 65536, 2)):
                if (lv, rw, bl, ns, ch) == (args.level, args.rows, args.blocks, args.streams, args.channels):
                    continue
                try:
                    steps = max(20, args.steps // 3) if ns <= 1024 else 10
                    extra.append(side_measure(dev, capi, workload, lv, rw, bl, ns, steps, channels=ch, packed=not args.no_packed))
                    if ns > 1024:
                        extra[-1]["config"] = "BASELINE.json configs[4] at full size"
                except Exception as e:   # a side measurement must never sink the headline line
                    extra.append({"level": lv, "error": str(e)[:200]})
            try:
                extra.append(side_measure(dev, capi, workload, 0, 0, 0, 0, 20, verify=512, corpus_files=4000))
            except Exception as e:
                extra.append({"level": "corpus", "error": str(e)[:200]})
            out["other_levels_kernel_only"] = extra
            # ... and their key figures inside `roofline`, which the driver's record keeps whole (task 4): fraction of the 8 TB/s
            # roofline at 4 B/sample, the launch's duration from HIP events, and how many streams' PCM was CRC-compared with the oracle's
            oc = {}
            for e in extra:
                if "frac_hbm" not in e:
                    continue
                name = ("configs2_corpus_4000_files_levels7to9" if str(e.get("config", "")).startswith("BASELINE.json configs[2]") else
                        "configs4_65536_stereo_streams_level11_rows64" if e.get("config") else
                        "configs1_1024_streams_level7_rows16" if (e["level"], e["rows"], e["blocks"]) == (7, 16, 1000) else
                        "level%d_rows%d_%dstreams" % (e["level"], e["rows"], e["streams"]))
                oc[name] = {"frac": e["frac_hbm"], "launch_ms": e.get("launch_ms"), "verified_streams": e.get("verified_streams"),
                            "staged_form": e.get("staged_form"), "staged_bytes_per_sample": e.get("staged_bytes_per_sample"),
                            "frac_int16_form": (e.get("int16_form") or {}).get("frac_hbm")}
                if "samples_from_byteplane_form" in e:
                    oc[name]["samples_from_byteplane_form"] = e["samples_from_byteplane_form"]
            out["roofline"]["other_configs"] = oc
        if not args.no_extra and not args.no_cpu and len(batch.files) > 4:
            # informational: file bytes -> PCM in host memory through acm_batch_decode (bit parsing on the host pool
            # or on device lanes, PCIe both ways, pipelined); by contract this is NOT `value`
            try:
                files = [f.tobytes() for f in batch.files]
                e2e = {"streams": len(files), "host_threads": workload_cpus()}
                for name, mode, pin in (("host_parse", capi.PARSE_HOST, False), ("host_parse_int16_staging", capi.PARSE_HOST, False),
                                        ("host_parse_packed_staging", capi.PARSE_HOST, False),
                                        ("device_parse", capi.PARSE_DEVICE, False), ("device_parse_int16_staging", capi.PARSE_DEVICE, False),
                                        ("device_parse_pinned_out", capi.PARSE_DEVICE, True)):
                    # pinned_out: the caller's PCM buffers are pinned (acmhip_host_alloc), read-back lands in them directly.
                    # The hosts of this pool are shared: single calls show 1.5-2 x outliers in any mode (profiles/pinned_out_probe.py,
                    # VERDICT r3 Weak 5), so every leg is the best of three calls behind one that sizes the arenas, all totals kept
                    pkd = name == "host_parse_packed_staging"        # ACM_BATCH_STAGE_PACKED: the pool packs too, half the upload
                    # the library's default: the byte-plane form is written by the parsing pass itself - the host pool's (acm_stage_file_mform)
                    # or the device parser's column kernel - and the batch's plans read it on the chunk kernel (`packed_streams` = streams
                    # that travelled so).  *_int16_staging: ACM_BATCH_STAGE_INT16, every row as int16 (the default up to round 4)
                    bpl = False if name.endswith("_int16_staging") or pkd else None
                    capi.batch_decode(dev, files, threads=0, parse=mode, pinned=pin, packed=pkd, byteplane=bpl)
                    runs = []
                    for _ in range(1 if mode == capi.PARSE_HOST else 3):
                        res, tm = capi.batch_decode(dev, files, threads=0, parse=mode, pinned=pin, packed=pkd, byteplane=bpl)
                        runs.append((tm.total_s, tm))
                        del res
                    tm = min(runs, key=lambda r: r[0])[1]
                    e2e[name] = {"msamples_s": round(tm.samples / tm.total_s / 1e6, 1), "parse_s": round(tm.stage_s, 3),
                                 "h2d_s": round(tm.h2d_s, 3), "kernel_s": round(tm.kernel_s, 4), "d2h_s": round(tm.d2h_s, 3),
                                 "total_s": round(tm.total_s, 3), "device_parsed": tm.device_parsed, "h2d_bytes": tm.h2d_bytes,
                                 "packed_streams": tm.packed_streams,
                                 "total_s_every_call": [round(r[0], 3) for r in runs]}
                out["end_to_end"] = e2e
            except Exception as e:
                out["end_to_end"] = {"error": str(e)[:200]}
        if not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(batch, args.cpu_seconds)
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)[:200]}

    if rank == 0:
        # RCCL announces its version through C stdio when the first communicator comes up; that text sits in libc's buffer until
        # the process ends and would land BEHIND the line below - push it out first, so that the JSON line is the last line
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)

    plan.destroy()
    for p in pk_ptrs or ():
        dev.free(p)
    for p in d_mf or ():
        dev.free(p)
    if dist is not None:
        dev.free(bufs[0])
        dev.free(bufs[1])
    else:
        for p in bufs:
            dev.free(p)
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
