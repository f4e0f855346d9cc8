"""bench.py's launcher logic without a GPU: a bare `python bench.py --gpus N` must start its ranks itself, and when they
cannot run (this container has no GPU) it must fail loudly -- non-zero exit, no result line -- rather than print a number."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600, env=env)


def test_bare_multi_gpu_run_launches_ranks_and_fails_without_gpus():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for the GPU-less container")
    r = _run("--gpus", "2", "--debug-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launching 2 ranks" in r.stderr and "needs a GPU" in r.stderr
    r = _run("--gpus", "2", "--steps", "2")                    # RCCL backend: refuses before launching anything
    assert r.returncode != 0 and r.stdout.strip() == "" and "GPU(s) visible" in r.stderr


def test_single_gpu_run_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this check is for the GPU-less container")
    r = _run("--steps", "2", "--warmup", "1")
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs a GPU" in r.stderr
