"""Where the start-up time of a one-file decode goes (run on the GPU box): runtime init, device open, first launch."""
import sys, time
sys.path.insert(0, '.')
t0 = time.perf_counter()
import ctypes as C
from libacm_amd import _build
L = C.CDLL(_build.build_hip())           # no torch in this process: the system HIP runtime, as acmtool sees it
t1 = time.perf_counter()
L.acmhip_device_count.restype = C.c_int
n = L.acmhip_device_count()
t2 = time.perf_counter()
dev = C.c_void_p()
rc = L.acmhip_device_open(0, None, C.byref(dev))
t3 = time.perf_counter()
from libacm_amd import synth
f = synth.generate(seed=1, level=7, rows=16, nblocks=50)
import numpy as np
class Item(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("pcm", C.c_void_p), ("pcm_cap", C.c_size_t), ("words", C.c_uint64),
                ("status", C.c_int32), ("level", C.c_uint32), ("rows", C.c_uint32), ("channels", C.c_uint32), ("rate", C.c_uint32),
                ("total_values", C.c_uint32), ("reserved", C.c_uint32), ("dev_off", C.c_uint64)]
buf = np.frombuffer(f, dtype=np.uint8)
out = np.zeros(50 * 16 * 128, dtype=np.int16)
it = Item(buf.ctypes.data, buf.size, out.ctypes.data, out.size)
L.acm_batch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
ta = time.perf_counter()
p1 = C.c_void_p(); L.acmhip_malloc(dev, 1 << 20, C.byref(p1))
tb = time.perf_counter()
p2 = C.c_void_p(); L.acmhip_host_alloc(1 << 20, C.byref(p2))
tc = time.perf_counter()
L.acmhip_upload(dev, p1, p2, 1 << 20); L.acmhip_device_sync(dev)
td = time.perf_counter()
print("first hipMalloc %.3f s, first hipHostMalloc %.3f s, first copy+sync %.3f s" % (tb - ta, tc - tb, td - tc))
t4 = time.perf_counter()
rc1 = L.acm_batch_decode(dev, C.byref(it), 1, None, None)
t5 = time.perf_counter()
rc2 = L.acm_batch_decode(dev, C.byref(it), 1, None, None)
t6 = time.perf_counter()
print("dlopen %.3f s, device_count %.3f s, device_open %.3f s, first decode %.3f s (rc %d), second decode %.4f s (rc %d)"
      % (t1 - t0, t2 - t1, t3 - t2, t5 - t4, rc1, t6 - t5, rc2))
