"""ctypes binding of libacm_hip.so (include/acm_hip.h + include/libacm.h).

Thin plumbing for tests, bench.py and the multi-GPU front end.  Nothing here
computes: staging is done by the library's host parser, synthesis by its HIP
kernels.  There is no CPU synthesis fallback - without a usable HIP device
Device() raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

FMT_S16LE, FMT_S16BE, FMT_U16LE, FMT_U16BE = 0, 1, 2, 3
PLAN_AUTO, PLAN_STAGEWISE = 0, 1
PLAN_FORM_ONLY, PLAN_UPLOAD_ASYNC = 0x100, 0x200          # modifiers, or-ed in (acm_hip.h)
PLAN_LEAN_ALWAYS, PLAN_NO_LEAN, PLAN_FORCE_HALO, PLAN_FORCE_CARRY = 0x400, 0x800, 0x1000, 0x2000
# test plumbing: or-ed into the flags of every Plan / batch call made through this module (fixtures that used to set ACM_K2, ACM_K1_CARRY,
# ACM_BATCH_RANGES in the environment set these instead: the shipped library reads no kernel-selection switch from the environment)
PLAN_EXTRA = 0
BATCH_EXTRA = 0


def batch_ranges(n):
    """ACM_BATCH_RANGES(n) of acm_hip.h: device parsing in n block ranges (1 = in one piece, 0 = the library decides)"""
    return (int(n) & 0xFF) << 8
ERR_NO_DEVICE = -101
ERR_ARG = -103
ERR_RANGE = -105


class BlkHdr(C.Structure):
    _fields_ = [("val", C.c_uint32), ("pwr", C.c_uint32)]


class Patch(C.Structure):
    _fields_ = [("sample", C.c_uint64), ("value", C.c_int32), ("stream", C.c_uint32)]


class StreamDesc(C.Structure):
    _fields_ = [("idx_off", C.c_uint64), ("hdr_off", C.c_uint64), ("pcm_off", C.c_uint64),
                ("n_emit", C.c_uint64), ("level", C.c_uint32), ("rows", C.c_uint32),
                ("nrows", C.c_uint32), ("row_begin", C.c_uint32)]


class PlanStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("tiles", C.c_uint64), ("fused_streams", C.c_uint32),
                ("stagewise_streams", C.c_uint32), ("launches", C.c_uint32), ("reserved", C.c_uint32),
                ("mform_tiles", C.c_uint32), ("packed_tiles", C.c_uint32)]


class PackedChunk(C.Structure):
    _fields_ = [("blob_off16", C.c_uint32), ("count", C.c_uint16), ("kind", C.c_uint8), ("row0", C.c_uint8)]


class PackedStream(C.Structure):
    _fields_ = [("chunk_off", C.c_uint64), ("ntiles", C.c_uint32), ("form", C.c_uint32)]


FORM_PACKED, FORM_BYTEPLANE = 0, 1


PACKED_CHUNK_DT = np.dtype([("blob_off16", "<u4"), ("count", "<u2"), ("kind", "u1"), ("row0", "u1")])


class StageInfo(C.Structure):
    _fields_ = [("level", C.c_uint32), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("channels", C.c_uint32), ("hdr_channels", C.c_uint32), ("rate", C.c_uint32),
                ("total_values", C.c_uint32), ("wavc", C.c_uint32), ("blocks", C.c_uint32),
                ("end_status", C.c_int32), ("npatches", C.c_uint64), ("header_bytes", C.c_uint64)]


class BatchItem(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("pcm", C.c_void_p), ("pcm_cap", C.c_size_t),
                ("words", C.c_uint64), ("status", C.c_int32), ("level", C.c_uint32), ("rows", C.c_uint32),
                ("channels", C.c_uint32), ("rate", C.c_uint32), ("total_values", C.c_uint32), ("reserved", C.c_uint32),
                ("dev_off", C.c_uint64)]


class BatchOpts(C.Structure):
    _fields_ = [("force_chans", C.c_int), ("fmt", C.c_uint), ("threads", C.c_int), ("plan_flags", C.c_uint),
                ("parse", C.c_uint), ("flags", C.c_uint), ("d_pcm", C.c_void_p), ("d_pcm_words", C.c_uint64),
                ("prestaged", C.c_void_p)]


class BatchTiming(C.Structure):
    _fields_ = [("stage_s", C.c_double), ("h2d_s", C.c_double), ("kernel_s", C.c_double),
                ("d2h_s", C.c_double), ("total_s", C.c_double), ("samples", C.c_uint64), ("alloc_s", C.c_double),
                ("device_parsed", C.c_uint64), ("host_parsed", C.c_uint64), ("packed_streams", C.c_uint64), ("h2d_bytes", C.c_uint64)]


# every symbol include/acm_hip.h declares (checked by tests/test_abi.py)
ACMHIP_SYMBOLS = [
    "acmhip_last_error", "acmhip_device_count", "acmhip_device_open", "acmhip_device_close",
    "acmhip_device_sync", "acmhip_device_stream", "acmhip_malloc", "acmhip_free", "acmhip_host_alloc",
    "acmhip_host_free", "acmhip_upload", "acmhip_download", "acmhip_memset", "acmhip_host_synth", "acmhip_set_host_synth_limit", "acmhip_host_synth_limit", "acmhip_set_seek_index", "acmhip_plan_create", "acmhip_plan_destroy",
    "acmhip_plan_launch", "acmhip_plan_get_stats", "acmhip_plan_form_rows", "acmhip_plan_time", "acm_stage_probe", "acm_stage_file", "acm_stage_file_mform",
    "acm_batch_decode", "acm_batch_pcm_words", "acm_batch_prestage", "acm_batch_prestage_free", "acmhip_prewarm",
    "acmhip_packed_tile_rows", "acmhip_packed_group_rows", "acmhip_packed_slots", "acmhip_pack_bound", "acmhip_pack_tiles", "acmhip_unpack_tile",
    "acmhip_plan_create_packed", "acmhip_plan_bind_packed",
    "acmhip_mform_tile_rows", "acmhip_mform_group", "acmhip_mform_bytes", "acmhip_mform_pairs", "acmhip_mform_rows", "acmhip_mform_unrows", "acmhip_plan_bind_mform",
]
# the 19 entry points of include/libacm.h (reference src/libacm.h:120-170)
LIBACM_SYMBOLS = [
    "acm_open_decoder", "acm_read", "acm_close", "acm_open_file", "acm_info", "acm_seekable", "acm_bitrate",
    "acm_rate", "acm_channels", "acm_raw_total", "acm_raw_tell", "acm_pcm_total", "acm_pcm_tell",
    "acm_time_total", "acm_time_tell", "acm_read_loop", "acm_seek_pcm", "acm_seek_time", "acm_strerror",
]

_lib = None


def lib_path():
    return os.path.join(_build.LIB, "libacm_hip.so")


def lib():
    """Load libacm_hip.so (building it if the tree is newer)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7; if the system copy
    # gets loaded first (through libacm_hip.so's DT_NEEDED) and torch later brings its own, the second one
    # finds no GPUs.  Importing torch first makes the loader resolve our DT_NEEDED to the copy already mapped.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    path = os.environ.get("ACM_HIP_LIB") or _build.build_hip()      # ACM_HIP_LIB: A/B runs of experimental builds
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    L.acmhip_last_error.restype = C.c_char_p
    L.acmhip_device_open.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.acmhip_device_close.argtypes = [vp]
    L.acmhip_device_close.restype = None
    L.acmhip_device_sync.argtypes = [vp]
    L.acmhip_device_stream.argtypes = [vp]
    L.acmhip_device_stream.restype = vp
    L.acmhip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.acmhip_free.argtypes = [vp, vp]
    L.acmhip_host_alloc.argtypes = [sz, C.POINTER(vp)]
    L.acmhip_host_free.argtypes = [vp]
    L.acmhip_upload.argtypes = [vp, vp, vp, sz]
    L.acmhip_download.argtypes = [vp, vp, vp, sz]
    L.acmhip_memset.argtypes = [vp, vp, C.c_int, sz]
    L.acmhip_host_synth.argtypes = [C.POINTER(StreamDesc), vp, vp, vp, sz, C.c_uint, vp]
    L.acmhip_set_host_synth_limit.argtypes = [C.c_uint64]
    L.acmhip_set_host_synth_limit.restype = None
    L.acmhip_host_synth_limit.restype = C.c_uint64
    L.acmhip_set_seek_index.argtypes = [C.c_int]
    L.acmhip_set_seek_index.restype = None
    L.acmhip_plan_create.argtypes = [vp, C.POINTER(StreamDesc), sz, C.POINTER(Patch), sz, C.c_uint, C.POINTER(vp)]
    L.acmhip_plan_destroy.argtypes = [vp]
    L.acmhip_plan_destroy.restype = None
    L.acmhip_plan_launch.argtypes = [vp, vp, vp, vp, C.c_uint]
    L.acmhip_plan_get_stats.argtypes = [vp, C.POINTER(PlanStats)]
    L.acmhip_plan_form_rows.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint64)]
    L.acmhip_plan_time.argtypes = [vp, vp, vp, vp, C.c_uint, C.c_int, C.POINTER(C.c_float)]
    L.acm_stage_probe.argtypes = [vp, sz, C.c_int, C.POINTER(StageInfo)]
    L.acm_stage_file.argtypes = [vp, sz, C.c_int, vp, vp, sz, vp, sz, C.POINTER(StageInfo)]
    L.acm_stage_file_mform.argtypes = [vp, sz, C.c_int, vp, vp, sz, C.POINTER(StageInfo), vp, C.c_uint64, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.acm_batch_decode.argtypes = [vp, C.POINTER(BatchItem), sz, C.POINTER(BatchOpts), C.POINTER(BatchTiming)]
    L.acm_batch_pcm_words.argtypes = [C.POINTER(BatchItem), sz, C.c_int]
    L.acm_batch_pcm_words.restype = C.c_uint64
    L.acm_batch_prestage.argtypes = [C.POINTER(BatchItem), sz, C.POINTER(BatchOpts), C.POINTER(vp), C.POINTER(C.c_double)]
    L.acm_batch_prestage_free.argtypes = [vp]
    L.acm_batch_prestage_free.restype = None
    L.acmhip_packed_tile_rows.argtypes = [C.c_uint32]
    L.acmhip_packed_group_rows.argtypes = [C.c_uint32]
    L.acmhip_packed_slots.argtypes = [C.c_uint32]
    L.acmhip_pack_bound.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    L.acmhip_pack_tiles.argtypes = [C.c_uint32, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.acmhip_unpack_tile.argtypes = [C.c_uint32, vp, vp, vp]
    L.acmhip_plan_create_packed.argtypes = [vp, C.POINTER(StreamDesc), sz, C.POINTER(PackedStream), C.POINTER(Patch), sz, C.c_uint, C.POINTER(vp)]
    L.acmhip_plan_bind_packed.argtypes = [vp, vp, vp]
    L.acmhip_mform_tile_rows.argtypes = [C.c_uint32]
    L.acmhip_mform_group.argtypes = [C.c_uint32]
    L.acmhip_mform_bytes.argtypes = [C.c_uint32, C.c_uint64]
    L.acmhip_mform_bytes.restype = C.c_uint64
    L.acmhip_mform_pairs.argtypes = [C.c_uint64]
    L.acmhip_mform_pairs.restype = C.c_uint64
    L.acmhip_mform_rows.argtypes = [C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, vp, C.POINTER(C.c_uint64)]
    L.acmhip_mform_unrows.argtypes = [C.c_uint32, vp, vp, C.c_uint64, vp]
    L.acmhip_plan_bind_mform.argtypes = [vp, vp, vp]
    _lib = L
    return L


class AcmHipError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise AcmHipError("%s failed (%d): %s" % (what, rc, lib().acmhip_last_error().decode(errors="replace")))


def device_count():
    return lib().acmhip_device_count()


# --------------------------------------------------------------------------- host staging
class Staged:
    """One file in staged form: idx (int16, PCM order), hdr (uint32[blocks,2] = val,pwr), patches, info."""

    def __init__(self, idx, hdr, patches, info):
        self.idx, self.hdr, self.patches, self.info = idx, hdr, patches, info

    @property
    def block_len(self):
        return self.info.rows * self.info.cols

    @property
    def words(self):
        """words an acm_read_loop() caller would get (whole blocks, total_values cut, channel rounding)"""
        pos, bl, tot, ch = 0, self.block_len, self.info.total_values, self.info.channels
        for _ in range(self.info.blocks):
            if pos >= tot:
                break
            take = min(bl, tot - pos)
            if ch > 1:
                take -= take % ch
            pos += take
            if take != bl:
                break
        return pos


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def probe(data, force_chans=0):
    a = _as_u8(data)
    info = StageInfo()
    rc = lib().acm_stage_probe(a.ctypes.data, a.size, force_chans, C.byref(info))
    return rc, info


def stage_file(data, force_chans=0, idx_out=None, hdr_out=None):
    """Host bit parsing of a whole file image -> Staged (raises on a non-ACM file)."""
    a = _as_u8(data)
    rc, info = probe(a, force_chans)
    if rc != 0:
        raise ValueError("not an ACM stream (%d)" % rc)
    bl = info.rows * info.cols
    need = (info.total_values + bl - 1) // bl
    idx = idx_out if idx_out is not None else np.zeros(need * bl, dtype=np.int16)
    hdr = hdr_out if hdr_out is not None else np.zeros((need, 2), dtype=np.uint32)
    assert idx.size >= need * bl and hdr.shape[0] >= need
    info2 = StageInfo()
    rc = lib().acm_stage_file(a.ctypes.data, a.size, force_chans, idx.ctypes.data, hdr.ctypes.data, need,
                              None, 0, C.byref(info2))
    if rc != 0:
        raise ValueError("acm_stage_file failed (%d)" % rc)
    patches = None
    if info2.npatches:
        patches = (Patch * info2.npatches)()
        rc = lib().acm_stage_file(a.ctypes.data, a.size, force_chans, idx.ctypes.data, hdr.ctypes.data, need,
                                  patches, info2.npatches, C.byref(info2))
    return Staged(idx[:info2.blocks * bl], hdr[:info2.blocks], patches, info2)


def stage_file_mform(data, force_chans=0, mf_base=0):
    """acm_stage_file_mform: the byte-plane form written by the parsing pass itself.  Returns (info, idx, hdr, blob, pairs, mf_rows, mf_bytes);
    idx is filled with -12345 beforehand, so the rows the call leaves alone show."""
    a = _as_u8(data)
    rc, info = probe(a, force_chans)
    if rc != 0:
        raise ValueError("not an ACM stream (%d)" % rc)
    bl = info.rows * info.cols
    need = (info.total_values + bl - 1) // bl
    idx = np.full(need * bl, -12345, dtype=np.int16)
    hdr = np.zeros((need, 2), dtype=np.uint32)
    nrows = (need * info.rows) & ~1
    blob = np.zeros(int(lib().acmhip_mform_bytes(info.level, nrows)) + 256, dtype=np.uint8)
    pairs = np.zeros(int(lib().acmhip_mform_pairs(nrows)) + 32, dtype=np.uint32)
    info2 = StageInfo()
    rows, nbytes = C.c_uint64(), C.c_uint64()
    rc = lib().acm_stage_file_mform(a.ctypes.data, a.size, force_chans, idx.ctypes.data, hdr.ctypes.data, need, C.byref(info2),
                                    blob.ctypes.data, mf_base, pairs.ctypes.data, C.byref(rows), C.byref(nbytes))
    if rc != 0:
        raise ValueError("acm_stage_file_mform failed (%d)" % rc)
    return info2, idx, hdr, blob, pairs, rows.value, nbytes.value


# --------------------------------------------------------------------------- host staging, packed half
def packed_tile_rows(level):
    return lib().acmhip_packed_tile_rows(level)


class PackedArena:
    """Host tables of the packed staged form of several streams: .chunks (acmhip_packed_slots(level) entries per tile), .blob,
    and .streams (PackedStream beside each stream descriptor)."""

    def __init__(self, chunks, blob, streams):
        self.chunks, self.blob, self.streams = chunks, blob, streams

    def upload(self, dev):
        ptrs = []
        for a in (self.chunks, self.blob):
            p = dev.malloc(max(a.nbytes, 16))
            flat = a.view(np.uint8).reshape(-1)
            for o in range(0, flat.size, 1 << 28):
                dev.upload(p + o, flat[o:o + (1 << 28)])
            ptrs.append(p)
        return tuple(ptrs)

    @property
    def nbytes(self):
        return self.chunks.nbytes + self.blob.nbytes


class MformArena:
    """The byte-plane staged form (acmhip_mform_rows) of several streams: .data (uint8 arena), .pairs (uint32 pair table: offset in
    64-byte units << 2 | width class) and .streams (PackedStream with form = FORM_BYTEPLANE beside each stream descriptor)."""

    def __init__(self, data, pairs, streams):
        self.data, self.pairs, self.streams = data, pairs, streams

    def upload(self, dev):
        ptrs = []
        for a in (self.data, self.pairs):
            p = dev.malloc(max(a.nbytes, 16))
            flat = a.view(np.uint8).reshape(-1)
            for o in range(0, flat.size, 1 << 28):
                dev.upload(p + o, flat[o:o + (1 << 28)])
            ptrs.append(p)
        return tuple(ptrs)

    @property
    def nbytes(self):
        return self.data.nbytes + self.pairs.nbytes

    def class_counts(self):
        """pair-table entries per class code: 1 = 12 bits at levels 8-12 (4 bits in level 7's form), 2 = 8 bits, 3 = 16 bits as two signed
        bytes; 0 = 16 bits over the whole int16 range at levels 8-12 - and the table's 32 entries of read slack, which are zeros"""
        return np.bincount(self.pairs & 3, minlength=4)


def mform_streams(idx, descs, threads=1, L=None):
    """The byte-plane form of every stream of a staged arena that can have one (whole tiles from row 0 of the levels
    acmhip_mform_tile_rows() covers).  L: another build of the library to stage with (profiles/ab_kernels.py --own-form)."""
    from concurrent.futures import ThreadPoolExecutor
    L = L or lib()
    n = len(descs)
    ntiles, rows, cap, p_at = [0] * n, [0] * n, [0] * n, [0] * n
    np_tot = 0
    for i, d in enumerate(descs):
        tr = L.acmhip_mform_tile_rows(d.level)
        if tr > 0 and d.row_begin == 0:
            ntiles[i] = min(d.nrows, d.n_emit >> d.level) // tr
            if (ntiles[i] * tr) & 1:            # the form is written pair by pair (chunks of one row: an even number of them)
                ntiles[i] -= 1
            rows[i] = ntiles[i] * tr
        p_at[i] = np_tot
        if ntiles[i]:
            cap[i] = (L.acmhip_mform_bytes(d.level, rows[i]) + 255) // 256 * 256
            np_tot += L.acmhip_mform_pairs(rows[i])
    pairs = np.zeros(np_tot + 32, dtype=np.uint32)             # (the kernel's scalar loads fetch whole groups of entries)
    parts = [None] * n

    def one(i):
        if ntiles[i]:
            d = descs[i]
            buf = np.empty(cap[i], dtype=np.uint8)
            used = C.c_uint64()
            rc = L.acmhip_mform_rows(d.level, idx[d.idx_off:].ctypes.data, rows[i], buf.ctypes.data, 0, pairs[p_at[i]:].ctypes.data,
                                     C.byref(used))
            if rc == ERR_RANGE:                 # (libraries of rounds 5 / 6, A/B runs: an index the form could not hold - the stream stays int16)
                ntiles[i] = 0
                return
            _check(rc, "acmhip_mform_rows")
            parts[i] = buf[:(used.value + 63) // 64 * 64]
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(one, range(n)))
    total = sum(p.size for p in parts if p is not None)
    if (total >> 6) >= 1 << 30:
        raise AcmHipError("byte-plane arena beyond 64 GB")
    at = 0
    for i in range(n):
        if parts[i] is not None:
            k = int(L.acmhip_mform_pairs(rows[i]))
            pairs[p_at[i]:p_at[i] + k] += np.uint32((at // 64) << 2)        # offsets were written relative to the stream's own block
            at += parts[i].size
    data = np.empty(max(at, 16) + 64, dtype=np.uint8)
    at = 0
    for i in range(n):
        if parts[i] is not None:
            data[at:at + parts[i].size] = parts[i]
            at += parts[i].size
    data[at:] = 0
    return MformArena(data, pairs, [PackedStream(p_at[i], ntiles[i], FORM_BYTEPLANE) for i in range(n)])


def mform_unrows(level, blob, pairs, nrows):
    out = np.zeros(nrows << level, dtype=np.int16)
    _check(lib().acmhip_mform_unrows(level, blob.ctypes.data, pairs.ctypes.data, nrows, out.ctypes.data), "acmhip_mform_unrows")
    return out


def pack_streams(idx, descs, threads=1):
    """The packed staged form of every stream of a staged arena that can have one (whole tiles from row 0 of the levels
    acmhip_packed_tile_rows() covers): returns a PackedArena whose .streams is a list of PackedStream beside descs."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    n = len(descs)
    ntiles, chunk_off, blob_cap, slots = [0] * n, [0] * n, [0] * n, [0] * n
    c_at = 0
    for i, d in enumerate(descs):
        tr = L.acmhip_packed_tile_rows(d.level)
        if tr > 0 and d.row_begin == 0:
            ntiles[i] = min(d.nrows, d.n_emit >> d.level) // tr
            slots[i] = L.acmhip_packed_slots(d.level)
        chunk_off[i] = c_at
        c_at += ntiles[i] * slots[i]
        if ntiles[i]:
            mb = C.c_uint64()
            _check(L.acmhip_pack_bound(d.level, ntiles[i], C.byref(mb)), "acmhip_pack_bound")
            blob_cap[i] = (mb.value + 15) // 16 * 16
    chunks = np.zeros(max(c_at, 1), dtype=PACKED_CHUNK_DT)
    parts = [None] * n

    def one(i):
        if not ntiles[i]:
            return
        d = descs[i]
        bl = np.zeros(blob_cap[i], dtype=np.uint8)
        nb = C.c_uint64()
        _check(L.acmhip_pack_tiles(d.level, idx[d.idx_off:].ctypes.data, ntiles[i], chunks[chunk_off[i]:].ctypes.data,
                                   bl.ctypes.data, 0, C.byref(nb)), "acmhip_pack_tiles")
        parts[i] = bl[:nb.value].copy()
    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(one, range(n)))
    b_at = 0
    lay = []
    for i in range(n):
        lay.append(b_at)
        if parts[i] is not None:
            b_at += (parts[i].size + 15) // 16 * 16
    blob = np.zeros(max(b_at, 16), dtype=np.uint8)
    for i in range(n):
        if parts[i] is None:
            continue
        b0 = lay[i]
        c = chunks[chunk_off[i]:chunk_off[i] + ntiles[i] * slots[i]]
        c["blob_off16"][c["kind"] != 0] += b0 // 16
        blob[b0:b0 + parts[i].size] = parts[i]
        parts[i] = None
    streams = [PackedStream(chunk_off[i], ntiles[i], 0) for i in range(n)]
    return PackedArena(chunks, blob, streams)


def unpack_tile(level, chunks, blob, entry):
    """inverse of the packer for the tile whose descriptors start at chunk-table entry `entry` (tests): int16 [tile_rows, cols]"""
    tr = lib().acmhip_packed_tile_rows(level)
    out = np.zeros((tr, 1 << level), dtype=np.int16)
    _check(lib().acmhip_unpack_tile(level, chunks[entry:].ctypes.data, blob.ctypes.data, out.ctypes.data), "acmhip_unpack_tile")
    return out


# --------------------------------------------------------------------------- device
class Device:
    def __init__(self, ordinal=0, hip_stream=None):
        self.h = C.c_void_p()
        rc = lib().acmhip_device_open(ordinal, hip_stream, C.byref(self.h))
        if rc != 0:
            self.h = None
            raise AcmHipError("acmhip_device_open(%d) failed (%d): %s"
                              % (ordinal, rc, lib().acmhip_last_error().decode(errors="replace")))

    def close(self):
        if self.h:
            lib().acmhip_device_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        _check(lib().acmhip_device_sync(self.h), "acmhip_device_sync")

    def malloc(self, nbytes):
        p = C.c_void_p()
        _check(lib().acmhip_malloc(self.h, nbytes, C.byref(p)), "acmhip_malloc")
        return p.value

    def free(self, ptr):
        if ptr:
            _check(lib().acmhip_free(self.h, ptr), "acmhip_free")

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().acmhip_upload(self.h, dptr, arr.ctypes.data, arr.nbytes), "acmhip_upload")
        self.sync()          # pageable numpy memory: finish before the array can go away

    def download(self, arr, dptr):
        _check(lib().acmhip_download(self.h, arr.ctypes.data, dptr, arr.nbytes), "acmhip_download")
        self.sync()

    def memset(self, dptr, byte, nbytes):
        """fill device memory (queued on the device stream); the parity tests poison a whole PCM arena with it before a launch"""
        _check(lib().acmhip_memset(self.h, dptr, byte, nbytes), "acmhip_memset")


class Plan:
    def __init__(self, dev, descs, patches=None, flags=PLAN_AUTO, packed=None):
        """packed: optional list of PackedStream, one per desc (ntiles 0 = that stream has no packed form)"""
        self.dev = dev
        flags |= PLAN_EXTRA
        n = len(descs)
        arr = (StreamDesc * max(n, 1))(*descs)
        np_ = len(patches) if patches is not None else 0
        self.h = C.c_void_p()
        if packed is not None:
            assert len(packed) == n
            pk = (PackedStream * max(n, 1))(*packed)
            _check(lib().acmhip_plan_create_packed(dev.h, arr, n, pk, patches if np_ else None, np_, flags, C.byref(self.h)),
                   "acmhip_plan_create_packed")
        else:
            _check(lib().acmhip_plan_create(dev.h, arr, n, patches if np_ else None, np_, flags, C.byref(self.h)),
                   "acmhip_plan_create")

    def bind_packed(self, d_chunks, d_blob):
        """device tables of the packed staged form for every later launch (both None: back to the int16 arena)"""
        _check(lib().acmhip_plan_bind_packed(self.h, d_chunks, d_blob), "acmhip_plan_bind_packed")

    def bind_mform(self, d_mform, d_pairs):
        """device arena and pair table of the byte-plane staged form for every later launch (both None: back to the int16 arena)"""
        _check(lib().acmhip_plan_bind_mform(self.h, d_mform, d_pairs), "acmhip_plan_bind_mform")

    def launch(self, d_idx, d_hdr, d_pcm, fmt=FMT_S16LE):
        _check(lib().acmhip_plan_launch(self.h, d_idx, d_hdr, d_pcm, fmt), "acmhip_plan_launch")

    def time(self, d_idx, d_hdr, d_pcm, fmt=FMT_S16LE, reps=1):
        ms = C.c_float()
        _check(lib().acmhip_plan_time(self.h, d_idx, d_hdr, d_pcm, fmt, reps, C.byref(ms)), "acmhip_plan_time")
        return ms.value

    def stats(self):
        st = PlanStats()
        _check(lib().acmhip_plan_get_stats(self.h, C.byref(st)), "acmhip_plan_get_stats")
        return st

    def form_rows(self, stream):
        """rows of stream `stream` this plan reads from the stream's second staged form (0: none of them)"""
        rows = C.c_uint64()
        _check(lib().acmhip_plan_form_rows(self.h, stream, C.byref(rows)), "acmhip_plan_form_rows")
        return rows.value

    def destroy(self):
        if self.h:
            # a plan that outlived its device handle (a test that failed before its destroy(), collected at interpreter exit) is the
            # handle's memory gone with it: nothing left to hand back, and the handle must not be touched
            if getattr(self.dev, "h", None):
                lib().acmhip_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _round_up(v, a):
    return (v + a - 1) // a * a


class Arena:
    """Several staged streams laid out in three host arrays + their descriptors (whole-stream decode)."""

    def __init__(self, staged_list, windows=None):
        """windows: optional per-stream (row_begin, n_emit) to decode only a slice"""
        idx_tot = hdr_tot = pcm_tot = 0
        self.descs, self.patch_list = [], []
        lay = []
        for i, s in enumerate(staged_list):
            bl = s.block_len
            n = s.info.blocks * bl
            row_begin, n_emit = (0, s.words) if windows is None else windows[i]
            d = StreamDesc(idx_off=idx_tot, hdr_off=hdr_tot, pcm_off=pcm_tot, n_emit=n_emit,
                           level=s.info.level, rows=s.info.rows, nrows=s.info.blocks * s.info.rows,
                           row_begin=row_begin)
            self.descs.append(d)
            lay.append((idx_tot, hdr_tot, pcm_tot, n, n_emit))
            if s.patches is not None:
                for p in s.patches:
                    self.patch_list.append(Patch(p.sample, p.value, i))
            idx_tot += _round_up(max(n, 1), 64)
            hdr_tot += max(s.info.blocks, 1)
            pcm_tot += _round_up(max(n_emit, 1), 64)
        self.idx = np.zeros(idx_tot, dtype=np.int16)
        self.hdr = np.zeros((hdr_tot, 2), dtype=np.uint32)
        self.pcm_words = pcm_tot
        self.layout = lay
        for s, (io, ho, po, n, ne) in zip(staged_list, lay):
            self.idx[io:io + n] = s.idx[:n]
            self.hdr[ho:ho + s.info.blocks] = s.hdr[:s.info.blocks]
        self.patches = (Patch * len(self.patch_list))(*self.patch_list) if self.patch_list else None


def synth(dev, staged_list, fmt=FMT_S16LE, flags=PLAN_AUTO, windows=None, return_stats=False, patch_subset=None, packed=False, mform=False):
    """Upload staged streams, run the hot path once, return one PCM array (uint16 view of the bytes) per stream.
    patch_subset (tests): keep only these entries of the batch's H1 patch list.
    packed: stage the packed form too (acmhip_pack_tiles) and bind it: whole tiles from row 0 are read from it.
    mform: the same with the byte-plane form (acmhip_mform_rows) and the matrix-core build of the lean kernel."""
    ar = Arena(staged_list, windows)
    if patch_subset is not None and ar.patch_list:
        ar.patch_list = [ar.patch_list[k] for k in patch_subset]
        ar.patches = (Patch * len(ar.patch_list))(*ar.patch_list) if ar.patch_list else None
    d_idx = dev.malloc(ar.idx.nbytes)
    d_hdr = dev.malloc(ar.hdr.nbytes)
    d_pcm = dev.malloc(ar.pcm_words * 2)
    try:
        dev.upload(d_idx, ar.idx)
        dev.upload(d_hdr, ar.hdr)
        pk, pk_ptrs = None, ()
        if packed:
            patched = {p.stream for p in ar.patch_list}
            pk = pack_streams(ar.idx, ar.descs)
            for i in patched:
                pk.streams[i].ntiles = 0
            pk_ptrs = pk.upload(dev)
        elif mform:
            patched = {p.stream for p in ar.patch_list}
            pk = mform_streams(ar.idx, ar.descs)
            for i in patched:
                pk.streams[i].ntiles = 0
            pk_ptrs = pk.upload(dev)
        plan = Plan(dev, ar.descs, ar.patches, flags, packed=pk.streams if pk else None)
        if pk and mform:
            plan.bind_mform(*pk_ptrs)
        elif pk:
            plan.bind_packed(*pk_ptrs)
        plan.launch(d_idx, d_hdr, d_pcm, fmt)
        out = np.zeros(ar.pcm_words, dtype=np.uint16)
        dev.download(out, d_pcm)
        st = plan.stats()
        plan.destroy()
        for p in pk_ptrs:
            dev.free(p)
    finally:
        dev.free(d_idx)
        dev.free(d_hdr)
        dev.free(d_pcm)
    res = [out[po:po + ne].copy() for (_, _, po, _, ne) in ar.layout]
    return (res, st) if return_stats else res


PARSE_HOST, PARSE_DEVICE, PARSE_AUTO = 0, 1, 2


BATCH_PCM_PINNED = 1
BATCH_STAGE_PACKED = 2
BATCH_STAGE_BYTEPLANE = 4
BATCH_STAGE_INT16 = 8


def batch_decode(dev, files, force_chans=0, fmt=FMT_S16LE, threads=0, flags=PLAN_AUTO, parse=PARSE_HOST, pinned=False, prestage=False,
                 packed=False, byteplane=None):
    """acm_batch_decode over a list of bytes objects -> (list of (status, uint16 array), BatchTiming).

    pinned=True: the output buffers are carved from one pinned arena (acmhip_host_alloc) and the call is told so
    (ACM_BATCH_PCM_PINNED: the read-back engine writes them directly); the arrays returned are copies.
    prestage=True: the bit parsing runs first, on its own (acm_batch_prestage: no device involved), and the decode gets its result.
    packed=True: ACM_BATCH_STAGE_PACKED - the host pool also packs the whole tiles, the upload carries the packed form.
    byteplane: None = the library's default (the byte-plane form wherever a stream can have it: first pass on the matrix cores),
    True = ACM_BATCH_STAGE_BYTEPLANE spelled out, False = ACM_BATCH_STAGE_INT16 (every row staged as int16)."""
    n = len(files)
    bufs = [_as_u8(f) for f in files]
    infos = [probe(b, force_chans) for b in bufs]
    sizes = [i.total_values if rc == 0 else 0 for rc, i in infos]
    arena = C.c_void_p()
    if pinned:
        offs, at = [], 0
        for sz in sizes:
            offs.append(at)
            at += (sz + 63) // 64 * 64
        _check(lib().acmhip_host_alloc(max(at, 1) * 2, C.byref(arena)), "acmhip_host_alloc")
        whole = np.ctypeslib.as_array(C.cast(arena, C.POINTER(C.c_uint16)), shape=(max(at, 1),))
        whole[:] = 0
        outs = [whole[o:o + sz] for o, sz in zip(offs, sizes)]
    else:
        # resident buffers (np.zeros hands out untouched pages: the library's copy-out threads would spend the call
        # taking first-touch page faults, which is the caller's allocation policy, not the decode)
        outs = [np.empty(sz, dtype=np.uint16) for sz in sizes]
        for o in outs:
            o.fill(0)
    items = (BatchItem * max(n, 1))()
    for k in range(n):
        items[k].data = bufs[k].ctypes.data
        items[k].len = bufs[k].size
        items[k].pcm = outs[k].ctypes.data if outs[k].size else None
        items[k].pcm_cap = outs[k].size
    opts = BatchOpts(force_chans, fmt, threads, flags | PLAN_EXTRA, parse, BATCH_EXTRA | (BATCH_PCM_PINNED if pinned else 0) | (BATCH_STAGE_PACKED if packed else 0) |
                     (BATCH_STAGE_BYTEPLANE if byteplane else 0) | (BATCH_STAGE_INT16 if byteplane is False else 0))
    tm = BatchTiming()
    pre = C.c_void_p()
    try:
        if prestage:
            _check(lib().acm_batch_prestage(items, n, C.byref(opts), C.byref(pre), None), "acm_batch_prestage")
            opts.prestaged = pre
        _check(lib().acm_batch_decode(dev.h, items, n, C.byref(opts), C.byref(tm)), "acm_batch_decode")
        res = [(items[k].status, outs[k][:items[k].words].copy() if pinned else outs[k][:items[k].words]) for k in range(n)]
    finally:
        if pre:
            lib().acm_batch_prestage_free(pre)
        if pinned:
            del outs, whole
            lib().acmhip_host_free(arena)
    return res, tm


def _batch_items(files):
    bufs = [_as_u8(f) for f in files]
    items = (BatchItem * max(len(files), 1))()
    for k, b in enumerate(bufs):
        items[k].data = b.ctypes.data
        items[k].len = b.size
    return bufs, items


def batch_pcm_words(files, force_chans=0):
    """16-bit words of device memory batch_decode_device needs for these files."""
    bufs, items = _batch_items(files)
    return int(lib().acm_batch_pcm_words(items, len(files), force_chans))


def batch_decode_device(dev, files, d_pcm, d_pcm_words, force_chans=0, fmt=FMT_S16LE, threads=0, flags=PLAN_AUTO,
                        parse=PARSE_AUTO):
    """acm_batch_decode with device-resident output: PCM of stream k lands at d_pcm + 2*offsets[k] bytes.

    Returns (statuses, words, offsets, BatchTiming); nothing is copied back to the host."""
    bufs, items = _batch_items(files)
    opts = BatchOpts(force_chans, fmt, threads, flags | PLAN_EXTRA, parse, BATCH_EXTRA, d_pcm, d_pcm_words)
    tm = BatchTiming()
    _check(lib().acm_batch_decode(dev.h, items, len(files), C.byref(opts), C.byref(tm)), "acm_batch_decode")
    n = len(files)
    return ([int(items[k].status) for k in range(n)], [int(items[k].words) for k in range(n)],
            [int(items[k].dev_off) for k in range(n)], tm)
