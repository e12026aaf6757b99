"""The whole N > 1 control flow of bench.py on the ONE GPU a test box has: `python bench.py --gpus 2 --backend gloo
--share-device` starts its own two ranks (torch.distributed.run), both on cuda:0 with a row shard each; pipelined steps
at B = 64 and B = 256, the packed candidate exchange (staged through the host around the gloo all-gather), the device key
merge, verification of the last timed batch against the synchronous sharded search, per-rank times gathered, ONE JSON line
on rank 0.  (RCCL itself needs one device per rank: that binding runs on a 1-rank group in tests/test_dropin_gpu.py and for
real the first time the driver has a multi-GPU node.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_one_gpu():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--rows", "300000",
                        "--steps", "8", "--warmup", "2"], capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                        # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["value"] > 0
    assert out["verified"]["last_pipelined_batch_equals_synchronous_search"] is True
    assert out["verified"]["batch256_last_pipelined_batch_equals_synchronous_search"] is True
    assert out["exchange_bindings"]["backend"] == "gloo" and out["exchange_bindings"]["torch_world_size"] == 2
    assert [p["rank"] for p in out["per_rank"]] == [0, 1] and sum(p["rows"] for p in out["per_rank"]) == 300000
    # 150 K-row shards are short scans: the passes alternate between two scan streams (no per-launch duration, a lifetime
    # instead)
    assert all((p[b]["kernel_ms"] or p[b]["kernel_lifetime_ms"]) > 0 for p in out["per_rank"] for b in ("batch64", "batch256"))
    assert out["roofline"]["two_scan_streams"] is True and out["roofline"]["kernel_ms"] is None
    assert out["extra"]["config3_batch256"]["last_pipelined_batch_equals_synchronous_search"] is True
    assert out["roofline"]["rows_per_gpu"] == 150000
    # rank 0 then ran the SAME workload through the single-process index (MultiDeviceIndex, what hooks.install builds for
    # num_shards = 2) in a child process: two logical shards on cuda:0 here
    sp = out["single_process"]
    assert "error" not in sp, sp
    assert sp["value"] > 0 and sp["verified"]["last_pipelined_batch_equals_synchronous_search"] is True
    assert sp["verified"]["batch256_last_pipelined_batch_equals_synchronous_search"] is True
    assert "one process, 2 shard(s)" in sp["process_model"] and sp["exchange"].startswith("host-mapped")


@pytest.mark.timeout(600)
def test_bench_single_process_mode_four_logical_shards():
    """`python bench.py --single-process --gpus 4 --share-device`: the JSON contract line from ONE process driving four row
    shards (MultiDeviceIndex) — no launcher, no collective."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "4", "--share-device", "--rows", "600000",
                        "--steps", "12", "--warmup", "2"], capture_output=True, text=True, timeout=550, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["steps"] == 12 and out["value"] > 0 and out["single_process"] is True
    assert out["config"]["shard_rows"] == [150000] * 4
    assert out["verified"] == {"last_pipelined_batch_equals_synchronous_search": True, "batch256_last_pipelined_batch_equals_synchronous_search": True}
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
