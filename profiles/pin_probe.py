#!/usr/bin/env python3
"""Why is the read-back into a freshly pinned 4 GB buffer sometimes half as fast?  (GPU box only)

Allocates the buffer again and again, two ways - hipHostMalloc (what acmhip_host_alloc did in round 3) and a 2 MB aligned
anonymous mapping with MADV_HUGEPAGE, touched, then hipHostRegister - and for each instance reports how much of it the kernel
backed with transparent huge pages (smaps AnonHugePages / ShmemPmdMapped, KernelPageSize) and the D2H / H2D rate into it.
"""
import ctypes as C
import mmap
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smaps_of(addr):
    """the smaps entry that contains addr -> dict of its kB fields"""
    out, hit = {}, False
    with open("/proc/self/smaps") as f:
        for line in f:
            m = re.match(r"^([0-9a-f]+)-([0-9a-f]+) ", line)
            if m:
                if hit:
                    break
                lo, hi = int(m.group(1), 16), int(m.group(2), 16)
                hit = lo <= addr < hi
                if hit:
                    out["range_mb"] = (hi - lo) >> 20
                    out["what"] = line.split(None, 5)[-1].strip() if len(line.split(None, 5)) > 5 else "anon"
                continue
            if hit:
                k, v = line.split(":", 1)
                if v.strip().endswith("kB"):
                    out[k] = int(v.split()[0])
    return out


def main():
    nbytes = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 4 << 30
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for p in ("enabled", "defrag", "shmem_enabled"):
        try:
            print("transparent_hugepage/%s: %s" % (p, open("/sys/kernel/mm/transparent_hugepage/" + p).read().strip()))
        except OSError:
            pass
    from libacm_amd import capi
    L = capi.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    hip.hipHostFree.argtypes = [C.c_void_p]
    libc = C.CDLL(None, use_errno=True)
    libc.mmap.restype = C.c_void_p
    libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
    libc.munmap.argtypes = [C.c_void_p, C.c_size_t]
    libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    dev = capi.Device(0)
    dptr = dev.malloc(nbytes)

    def rates(h):
        res = []
        for fn in (lambda: L.acmhip_download(dev.h, h, dptr, nbytes), lambda: L.acmhip_upload(dev.h, dptr, h, nbytes)):
            first = None
            best = 0
            for k in range(3):
                t0 = time.perf_counter()
                fn()
                dev.sync()
                r = nbytes / (time.perf_counter() - t0) / 1e9
                first = r if first is None else first
                best = max(best, r)
            res += [first, best]
        return res

    for rep in range(reps):
        for how in ("hipHostMalloc", "mmap+MADV_HUGEPAGE+hipHostRegister"):
            h = C.c_void_p()
            t0 = time.perf_counter()
            if how == "hipHostMalloc":
                rc = hip.hipHostMalloc(C.byref(h), nbytes, 0)
                base, span = h.value, nbytes
                C.memset(h, 1, nbytes)
            else:
                span = nbytes + (2 << 20)
                raw = libc.mmap(None, span, mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, -1, 0)
                base = raw
                al = (raw + (2 << 20) - 1) & ~((2 << 20) - 1)
                libc.madvise(al, nbytes, 14)            # MADV_HUGEPAGE
                C.memset(al, 1, nbytes)
                h = C.c_void_p(al)
                rc = hip.hipHostRegister(h, nbytes, 0)
            t_alloc = time.perf_counter() - t0
            if rc != 0:
                print("%s: failed (%d)" % (how, rc))
                continue
            sm = smaps_of(h.value)
            d2h_first, d2h_best, h2d_first, h2d_best = rates(h)
            print("%-36s alloc+touch %.2f s  mapping %5d MB %-12s AnonHuge %7d kB  ShmemPmd %7d kB  KernelPageSize %s kB | D2H first %5.1f best %5.1f  H2D first %5.1f best %5.1f GB/s"
                  % (how, t_alloc, sm.get("range_mb", -1), sm.get("what", "?")[:12], sm.get("AnonHugePages", -1), sm.get("ShmemPmdMapped", -1),
                     sm.get("KernelPageSize", "?"), d2h_first, d2h_best, h2d_first, h2d_best), flush=True)
            if how == "hipHostMalloc":
                hip.hipHostFree(h)
            else:
                hip.hipHostUnregister(h)
                libc.munmap(base, span)
    dev.free(dptr)
    dev.close()


if __name__ == "__main__":
    main()
