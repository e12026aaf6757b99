"""bench.py's own launcher (CPU tier): `python bench.py --gpus N` must start by itself — it re-executes under
torch.distributed.run — and must say so in its own words when the node has fewer GPUs than ranks were asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=env)


def test_multi_gpu_request_without_enough_gpus_fails_with_its_own_message():
    import torch
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box really has 8 GPUs")
    r = _run("--gpus", "8", "--steps", "1", "--warmup", "0", "--no-extra")
    assert r.returncode != 0
    assert "bench.py --gpus 8: this node shows" in r.stderr and "one rank per GPU" in r.stderr, r.stderr[-800:]
    assert "Traceback" not in r.stderr


def test_share_device_needs_gloo():
    import torch
    r = _run("--gpus", "2", "--share-device", "--steps", "1", "--warmup", "0", "--no-extra")
    assert r.returncode != 0
    want = "add --backend gloo" if torch.cuda.device_count() >= 1 else "needs one visible MI355X"
    assert want in r.stderr, r.stderr[-800:]
