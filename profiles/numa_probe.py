#!/usr/bin/env python3
"""Where does pinned host memory have to live for the read-back engine to run at full rate?  (GPU box only)

BENCH_r03 showed the device-parse batch's read-back at 0.166 s on the driver's box against 0.077 s on others
(Weak 5).  The boxes have two CPU sockets; a process is confined to 16 CPUs of one of them, which may or may
not be the socket the GPU hangs off.  This probe prints the topology and times D2H / H2D of 1 GiB for pinned buffers whose
pages were bound (set_mempolicy MPOL_BIND before hipHostMalloc) to each NUMA node in turn, and with the default policy.
"""
import ctypes as C
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MPOL_DEFAULT, MPOL_BIND = 0, 2
SYS_set_mempolicy = 238          # x86_64
libc = C.CDLL(None, use_errno=True)


def set_policy(node):
    if node is None:
        rc = libc.syscall(SYS_set_mempolicy, MPOL_DEFAULT, None, 0)
    else:
        mask = (C.c_ulong * 16)()
        mask[node // 64] = 1 << (node % 64)
        rc = libc.syscall(SYS_set_mempolicy, MPOL_BIND, mask, 1024)
    return rc, C.get_errno()


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def main():
    nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
    cpus = sorted(os.sched_getaffinity(0))
    print("NUMA nodes:", nodes)
    for n in nodes:
        print("  node %d cpus %s  mem %s" % (n, read("/sys/devices/system/node/node%d/cpulist" % n),
                                            (read("/sys/devices/system/node/node%d/meminfo" % n) or "").split("\n")[0].split(":")[-1].strip()))
    mine = {}
    for c in cpus:
        for p in glob.glob("/sys/devices/system/cpu/cpu%d/node*" % c):
            mine.setdefault(int(p.rsplit("node", 1)[1]), []).append(c)
    print("this process may run on %d cpus: %s" % (len(cpus), {k: "%d cpus" % len(v) for k, v in mine.items()}))
    print("cgroup cpu.max:", read("/sys/fs/cgroup/cpu.max"), " cpuset.mems.effective:", read("/sys/fs/cgroup/cpuset.mems.effective"))
    for d in sorted(glob.glob("/sys/class/drm/renderD*/device")):
        if read(d + "/vendor") == "0x1002":
            print("GPU %s numa_node %s local_cpulist %s" % (os.path.basename(os.path.realpath(d)), read(d + "/numa_node"), read(d + "/local_cpulist")))
    import numpy as np
    from libacm_amd import capi
    L = capi.lib()
    dev = capi.Device(0)
    nbytes = 1 << 30
    dptr = dev.malloc(nbytes)
    for node in [None] + nodes:
        rc, err = set_policy(node)
        if rc != 0:
            print("node %s: set_mempolicy failed (errno %d)" % (node, err))
            continue
        h = C.c_void_p()
        if L.acmhip_host_alloc(nbytes, C.byref(h)) != 0:
            print("node %s: pinned allocation failed: %s" % (node, L.acmhip_last_error().decode()))
            set_policy(None)
            continue
        C.memset(h, 1, nbytes)
        res = {}
        for name, fn in (("d2h", lambda: L.acmhip_download(dev.h, h, dptr, nbytes)), ("h2d", lambda: L.acmhip_upload(dev.h, dptr, h, nbytes))):
            best = 0
            for _ in range(4):
                t0 = time.perf_counter()
                fn()
                dev.sync()
                best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
            res[name] = best
        print("pinned pages on node %-7s  D2H %5.1f GB/s  H2D %5.1f GB/s" % ("default" if node is None else node, res["d2h"], res["h2d"]), flush=True)
        L.acmhip_host_free(h)
        set_policy(None)
    dev.free(dptr)
    dev.close()


if __name__ == "__main__":
    main()
