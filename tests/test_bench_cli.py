"""bench.py's launcher contract, checked without a GPU: `--gpus N` is never silently ignored."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_fails_loudly_when_devices_are_missing():
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert f"--gpus {n} requested but only" in r.stderr
    assert "{" not in r.stdout, "no bench line may be printed for a run that did not get its GPUs"


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2 but --gpus 1" in r.stderr
