"""Build recipes for the native pieces (in-tree, no JIT cache).

  libacm_amd/lib/libacm_hip.so   HIP kernels + C-ABI (include/acm_hip.h) + drop-in libacm API (include/libacm.h)
  libacm_amd/lib/libacmsynth.so  synthetic ACM writer (include/acm_synth.h), plain C
  libacm_amd/bin/acmtool         CLI clone linked against libacm_hip.so

hipcc cross-compiles for gfx950 without a GPU, so all of this builds in the
authoring container; the resulting files travel to the GPU box with the tree.
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "libacm_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
BIN = os.path.join(PKG, "bin")
INC = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
GFX = "gfx950"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _headers():
    return [os.path.join(INC, f) for f in os.listdir(INC)] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc"))]


def build_synth(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libacmsynth.so")
    src = [os.path.join(CSRC, "acm_synth.c")]
    if force or _stale(out, src + _headers()):
        _run(["gcc", "-O2", "-g", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", INC, "-o", out] + src)
    return out


HIP_SOURCES = ["acm_kernels.hip", "acm_parse.hip", "acm_hip_api.cpp", "acm_fill.cpp", "acm_pack.cpp", "acm_stream.cpp", "acm_batch.cpp",
               "acm_host_synth.cpp"]
HOST_ONLY = {"acm_host_synth.cpp"}      # plain C++ (CPU-dispatched AVX2 inside): no device pass


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libacm_hip.so")
    src = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
    if force or _stale(out, src + _headers()):
        objs = []
        for s in src:
            o = os.path.join(LIB, os.path.basename(s) + ".o")
            if force or _stale(o, [s] + _headers()):
                cmd = [HIPCC, "-O3", "-g1", "-std=c++17", "-fPIC", "-Wall", "-Wextra",
                       "--offload-arch=" + GFX, "-I", INC, "-I", CSRC, "-c", s, "-o", o]
                if os.environ.get("ACM_ABLATION"):      # timing-only kernel variants for profiling sessions
                    cmd.insert(1, "-DACM_ABLATION=1")
                if os.environ.get("ACM_TUNING"):        # the alternative tile geometries behind ACM_K1_VARIANT, environment switches live
                    cmd.insert(1, "-DACM_TUNING=1")
                extra = os.environ.get("ACM_HIPCC_EXTRA", "").split()           # compiler-flag experiments
                cmd[1:1] = extra
                if os.path.basename(s) in HOST_ONLY:
                    cmd = [c for c in cmd if not c.startswith("--offload-arch")]
                elif s.endswith(".cpp"):
                    cmd.insert(1, "-x")
                    cmd.insert(2, "hip")
                _run(cmd)
            objs.append(o)
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + GFX, "-o", out] + objs + ["-lpthread"])
    return out


def build_tuning(force=False):
    """libacm_amd/lib/exp/tuning.so: the same sources with -DACM_TUNING - the only build that reads kernel-selection switches (ACM_K3,
    ACM_K2, ACM_K1_VARIANT, ACM_PARSE_SCAN ...) from the environment.  Measurement scripts under profiles/ and the one test that keeps
    acm_tile2's matrix builds of levels 8-12 alive (tests/test_gpu_byteplane.py) load it through ACM_HIP_LIB; the product never does."""
    exp = os.path.join(LIB, "exp")
    os.makedirs(exp, exist_ok=True)
    out = os.path.join(exp, "tuning.so")
    src = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
    if force or _stale(out, src + _headers()):
        from concurrent.futures import ThreadPoolExecutor
        objs = [os.path.join(exp, "tuning." + os.path.basename(s) + ".o") for s in src]

        def one(k):
            s, o = src[k], objs[k]
            cmd = [HIPCC, "-DACM_TUNING=1", "-O3", "-g1", "-std=c++17", "-fPIC", "--offload-arch=" + GFX, "-I", INC, "-I", CSRC, "-c", s, "-o", o]
            if os.path.basename(s) in HOST_ONLY:
                cmd = [c for c in cmd if not c.startswith("--offload-arch")]
            elif s.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            _run(cmd)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(one, range(len(src))))
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + GFX, "-o", out] + objs + ["-lpthread"])
        for o in objs:
            os.remove(o)
    return out


def build_tools(force=False):
    os.makedirs(BIN, exist_ok=True)
    out = os.path.join(BIN, "acmtool")
    src = [os.path.join(CSRC, "acmtool.c")]
    lib = build_hip(force)
    if os.path.exists(src[0]) and (force or _stale(out, src + [lib] + _headers())):
        _run(["gcc", "-O2", "-g", "-Wall", "-Wextra", "-I", INC, "-o", out] + src +
             ["-L", LIB, "-lacm_hip", "-lpthread", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + LIB])
    return out


def build_bench_tools(force=False):
    """Measurement infrastructure (not the product): the copy kernels bench.py calls for the box's practical HBM ceiling."""
    src = os.path.join(ROOT, "profiles", "ubench", "copy_bw.hip")
    out = os.path.join(ROOT, "profiles", "ubench", "libcopybw.so")
    if os.path.exists(src) and (force or _stale(out, [src])):
        _run([HIPCC, "-O3", "-fPIC", "-shared", "-DCOPY_BW_LIB=1", "--offload-arch=" + GFX, "-o", out, src])
    return out


def build_oracle():
    """Test infrastructure: our CPU restatement, and (only where /root/reference exists) the real reference."""
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])


def build_all(force=False):
    build_synth(force)
    build_hip(force)
    build_tools(force)
    build_tuning(force)
    build_bench_tools(force)
    build_oracle()
