# configs[0] and friends on the GPU box: acmtool -d of one short / one long file against the reference tool (oracle/_ref/acmtool_ref)
cd "${GRAFT_REPO_ROOT:-.}"
python3 - <<'PY'
import subprocess, time, os, hashlib
from libacm_amd import synth
cases = {"c0_level7_2Msamples": dict(level=7, rows=16, nblocks=1000), "level9_6Msamples": dict(level=9, rows=16, nblocks=750), "level9_41Msamples": dict(level=9, rows=16, nblocks=5000), "level9_164Msamples": dict(level=9, rows=16, nblocks=20000)}
for name, kw in cases.items():
    f = synth.generate(seed=synth.BASE_SEED + 77, **kw)
    path = "/tmp/%s.acm" % name
    open(path, "wb").write(f)
    for tool, env in (("oracle/_ref/acmtool_ref", {}), ("libacm_amd/bin/acmtool", {}), ("libacm_amd/bin/acmtool", {"ACMTOOL_HOST_LIMIT": "0"}), ("libacm_amd/bin/acmtool", {"ACMTOOL_HOST_LIMIT": "4000000000"})):
        ts = []
        for _ in range(5):
            out = "/tmp/%s_%s.wav" % (name, os.path.basename(tool))
            t0 = time.perf_counter()
            r = subprocess.run([tool, "-d", "-q", "-o", out, path], capture_output=True, env=dict(os.environ, **env))
            ts.append(time.perf_counter() - t0)
        d = hashlib.sha256(open(out, "rb").read()).hexdigest()[:12]
        print("%-22s %-28s %-24s best %.4f s  median %.4f s  rc %d  wav %s" % (name, tool, env or "", min(ts), sorted(ts)[2], r.returncode, d), flush=True)
PY
