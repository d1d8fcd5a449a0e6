"""Argument validation and degenerate cases of the hot-path C ABI (include/acm_hip.h)."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_stream, oracle_pcm
from libacm_amd import capi

pytestmark = pytest.mark.gpu


def desc(**kw):
    d = dict(idx_off=0, hdr_off=0, pcm_off=0, n_emit=128, level=7, rows=1, nrows=1, row_begin=0)
    d.update(kw)
    return capi.StreamDesc(**d)


def test_rejects_bad_descriptors(dev):
    for bad in (dict(level=16), dict(rows=0), dict(rows=4096), dict(idx_off=4), dict(pcm_off=2),
                dict(row_begin=2), dict(n_emit=129), dict(nrows=2, row_begin=1, n_emit=129)):
        with pytest.raises(capi.AcmHipError):
            capi.Plan(dev, [desc(**bad)])
    p = capi.Patch(0, 1, 5)
    with pytest.raises(capi.AcmHipError):
        capi.Plan(dev, [desc()], (capi.Patch * 1)(p))
    assert capi.lib().acmhip_plan_create(None, None, 0, None, 0, 0, None) == -103
    h = C.c_void_p()
    assert capi.lib().acmhip_device_open(99, None, C.byref(h)) == -103


def test_empty_and_zero_emit_plans(dev):
    plan = capi.Plan(dev, [])
    plan.launch(None, None, None)
    st = plan.stats()
    assert (st.samples, st.launches) == (0, 0)
    plan.destroy()
    plan = capi.Plan(dev, [desc(n_emit=0), desc(n_emit=0, level=3)])
    plan.launch(None, None, None)
    assert plan.stats().launches == 0
    with pytest.raises(capi.AcmHipError):
        plan.launch(None, None, None, fmt=4)
    plan.destroy()


def test_plan_is_reusable_and_format_is_per_launch(dev):
    f = make_stream(4400, 8, 16, 30, cut=4)
    s = capi.stage_file(f)
    ar = capi.Arena([s])
    d_idx, d_hdr, d_pcm = dev.malloc(ar.idx.nbytes), dev.malloc(ar.hdr.nbytes), dev.malloc(ar.pcm_words * 2)
    dev.upload(d_idx, ar.idx)
    dev.upload(d_hdr, ar.hdr)
    plan = capi.Plan(dev, ar.descs)
    out = np.zeros(ar.pcm_words, dtype=np.uint16)
    for fmt in (0, 1, 2, 3, 0):
        plan.launch(d_idx, d_hdr, d_pcm, fmt)
        dev.download(out, d_pcm)
        want, _ = oracle_pcm(f, 0, fmt & 1, 0 if fmt & 2 else 1)
        assert np.array_equal(out[:want.size], want)
    ms = plan.time(d_idx, d_hdr, d_pcm, reps=3)
    assert ms > 0
    plan.destroy()
    for p in (d_idx, d_hdr, d_pcm):
        dev.free(p)


def test_external_stream_handle(dev):
    """a caller-owned hipStream_t (here torch's) can carry the launches"""
    import torch
    st = torch.cuda.Stream()
    with capi.Device(0, st.cuda_stream) as d2:
        assert capi.lib().acmhip_device_stream(d2.h) == st.cuda_stream
        f = make_stream(4401, 7, 16, 25)
        got = capi.synth(d2, [capi.stage_file(f)])[0]
        assert np.array_equal(got, oracle_pcm(f)[0])


def test_more_streams_than_grid_y(dev):
    """stage-wise launches slice stream lists longer than gridDim.y (65535)"""
    f = make_stream(4402, 2, 3, 2)
    s = capi.stage_file(f)
    want, _ = oracle_pcm(f)
    n = 70000
    per = 64
    descs = [capi.StreamDesc(idx_off=0, hdr_off=0, pcm_off=k * per, n_emit=s.words, level=s.info.level,
                             rows=s.info.rows, nrows=s.info.blocks * s.info.rows, row_begin=0) for k in range(n)]
    idx = np.zeros(64, np.int16)
    idx[:s.idx.size] = s.idx
    d_idx, d_hdr, d_pcm = dev.malloc(idx.nbytes), dev.malloc(s.hdr.nbytes), dev.malloc(n * per * 2)
    dev.upload(d_idx, idx)
    dev.upload(d_hdr, s.hdr)
    plan = capi.Plan(dev, descs)
    plan.launch(d_idx, d_hdr, d_pcm)
    out = np.zeros(n * per, dtype=np.uint16)
    dev.download(out, d_pcm)
    out = out.reshape(n, per)[:, :want.size]
    assert np.array_equal(out, np.broadcast_to(want, out.shape))
    plan.destroy()
    for p in (d_idx, d_hdr, d_pcm):
        dev.free(p)


def test_launch_is_graph_capturable(dev):
    """acmhip_plan_launch neither allocates nor synchronises: a caller can capture it into a hipGraph
    (here through torch's CUDAGraph on torch's own stream) and replay it"""
    import torch
    f = make_stream(4403, 7, 16, 40)
    s = capi.stage_file(f)
    ar = capi.Arena([s])
    t_idx = torch.from_numpy(ar.idx).cuda()
    t_hdr = torch.from_numpy(ar.hdr.view(np.int32)).cuda()
    t_pcm = torch.zeros(ar.pcm_words, dtype=torch.int16, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        d2 = capi.Device(0, side.cuda_stream)
        plan = capi.Plan(d2, ar.descs)
        plan.launch(t_idx.data_ptr(), t_hdr.data_ptr(), t_pcm.data_ptr())      # warm (module load) outside capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            plan.launch(t_idx.data_ptr(), t_hdr.data_ptr(), t_pcm.data_ptr())
        t_pcm.zero_()
        g.replay()
        side.synchronize()
    want, _ = oracle_pcm(f)
    assert np.array_equal(t_pcm.cpu().numpy().view(np.uint16)[:want.size], want)
    plan.destroy()
    d2.close()


def test_kernel_selection_is_by_flag_not_by_environment(dev, monkeypatch):
    """what ACM_K2 / ACM_K1_CARRY used to switch through the environment are plan flags now (include/acm_hip.h): ACMHIP_PLAN_LEAN_ALWAYS
    puts a handful of whole tiles on the lean kernels (and so on the byte-plane form), ACMHIP_PLAN_NO_LEAN keeps everything off them,
    _FORCE_HALO / _FORCE_CARRY pick acm_fused_tile's flavour - same PCM every way - and the environment variables of earlier rounds change
    nothing in the shipped library"""
    files = [make_stream(61000 + i, lv, 16, 3 * (8192 >> lv) // 16 + 2, cut=i) for i, lv in enumerate((7, 8, 9, 10, 11))]
    staged = [capi.stage_file(f) for f in files]
    want = [oracle_pcm(f)[0] for f in files]
    seen = {}
    for name, flags in (("auto", capi.PLAN_AUTO), ("lean", capi.PLAN_LEAN_ALWAYS), ("no_lean", capi.PLAN_NO_LEAN),
                        ("halo", capi.PLAN_FORCE_HALO), ("carry", capi.PLAN_FORCE_CARRY), ("stagewise", capi.PLAN_STAGEWISE)):
        got, st = capi.synth(dev, staged, flags=flags, return_stats=True, mform=True)
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), name
        seen[name] = st.mform_tiles
    assert seen["lean"] > 0 and seen["no_lean"] == 0 and seen["stagewise"] == 0
    assert capi.lib().acmk_tuning_build() == 0
    for var, val in (("ACM_K2", "0"), ("ACM_K3", "0"), ("ACM_K1_CARRY", "1"), ("ACM_PREFIX", "0")):
        monkeypatch.setenv(var, val)
    got, st = capi.synth(dev, staged, flags=capi.PLAN_LEAN_ALWAYS, return_stats=True, mform=True)
    assert st.mform_tiles == seen["lean"] and all(np.array_equal(g, w) for g, w in zip(got, want))
