"""bench.py's launcher logic where no GPU is needed: it must fail loudly, never fall back (`--gpus` was inert)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def gpus_here():
    import torch
    return torch.cuda.device_count()


def test_more_gpus_than_visible_is_refused():
    n = gpus_here() + 1
    r = run_bench(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "%d GPUs requested, %d visible" % (n, n - 1) in r.stderr
    assert r.stdout.strip() == ""                   # no JSON line that could be mistaken for a measurement


def test_launcher_rank_count_must_match_the_flag():
    r = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"],
                  {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "--gpus 1 but the launcher started 2 rank(s)" in r.stderr


@pytest.mark.skipif(gpus_here() > 0, reason="needs a box without a GPU")
def test_no_gpu_no_number():
    r = run_bench(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "visible" in r.stderr or "no HIP device" in r.stderr
