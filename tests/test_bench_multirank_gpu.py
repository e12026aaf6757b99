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

from tools.bench_line import LINE_LIMIT, parse_emitted

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_side(stdout):
    """The final line (the ONLY stdout line that starts with `{`, <= 4 KB, contract keys) and the EXTRA record printed before it."""
    line, side = parse_emitted(stdout)
    last = [l for l in stdout.splitlines() if l.strip()][-1]
    assert last.startswith("{") and len(last) <= LINE_LIMIT and json.loads(last) == line
    assert set(line) == {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                         "data", "config", "roofline", "cpu_baseline", "verified"}
    assert line["data"] == "synthetic"
    return line, side



def _first_error_lines(stderr: str, n: int = 40) -> str:
    """What a crashed child said BEFORE its stack frames (the tail of a C++ abort is fifty `frame #` lines)."""
    keep = [ln for ln in stderr.splitlines() if not ln.startswith("frame #") and "amdgpu.ids" not in ln]
    return "\n".join(keep[:n])[:4000]

@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_one_gpu():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--rows", "300000",
                        "--steps", "8", "--warmup", "2"], capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], _first_error_lines(r.stderr), r.stderr[-3000:])
    out, side = _line_and_side(r.stdout)                            # ONE line, from rank 0
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["value"] > 0
    assert out["verified"]["last_pipelined_batch_equals_synchronous_search"] is True
    assert out["verified"]["batch256_last_pipelined_batch_equals_synchronous_search"] is True
    assert side["exchange_bindings"]["backend"] == "gloo" and side["exchange_bindings"]["torch_world_size"] == 2
    assert [p["rank"] for p in side["per_rank"]] == [0, 1] and sum(p["rows"] for p in side["per_rank"]) == 300000
    # 150 K-row shards are short scans: the passes alternate between two scan streams (no per-launch duration, a lifetime
    # instead)
    assert all((p[b]["kernel_ms"] or p[b]["kernel_lifetime_ms"]) > 0 for p in side["per_rank"] for b in ("batch64", "batch256"))
    assert out["roofline"]["two_scan_streams"] is True and out["roofline"]["kernel_ms"] is None
    assert side["extra"]["config3_batch256"]["last_pipelined_batch_equals_synchronous_search"] is True
    assert out["roofline"]["rows_per_gpu"] == 150000 and out["config"]["x_config3_batch256_qps"] > 0 and out["config"]["x_single_process_qps"] > 0
    # the encode half of the metric: every rank encoded its own chunks at the same time — per rank, and summed on the line
    rates = [p["corpus_embed_bf16"]["value"] for p in side["per_rank"]]
    assert all(r > 0 for r in rates) and out["config"]["x_corpus_embed_bf16_chunks_per_s_sum_over_ranks"] == pytest.approx(sum(rates), rel=1e-4)
    assert all(p["corpus_embed_bf16"]["encoder_path"] == "hip-fused-layers" and p["corpus_embed_bf16"]["gelu_path"] == "exact-erf-kernel" for p in side["per_rank"])
    # rank 0 then ran the SAME workload through the single-process index (MultiDeviceIndex, what hooks.install builds for
    # num_shards = 2) in a child process: two logical shards on cuda:0 here
    sp = side["single_process"]
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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "4", "--share-device", "--rows", "600000", "--steps", "12", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=270, env=env)
    if r.returncode < 0:
        # Round 6: ONE of this child's ~75 runs on the GPU pool died by SIGABRT inside PyTorch (an uncaught c10::Error; the test then kept only
        # the stack frames of its stderr) — in the full tier, directly behind the eight-rank test; 69 runs of the same command on their own and
        # five of this file passed.  A child killed by a signal is run once more, and what it said first is on record either way.
        import warnings
        warnings.warn(f"bench.py --single-process died by signal {-r.returncode}; first lines of its stderr:\n{_first_error_lines(r.stderr)}")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=270, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], _first_error_lines(r.stderr), r.stderr[-3000:])
    out, side = _line_and_side(r.stdout)
    assert out["n_gpus"] == 4 and out["steps"] == 12 and out["value"] > 0 and side["single_process"] is True
    assert out["config"]["shard_rows"] == [150000] * 4
    assert out["verified"] == {"last_pipelined_batch_equals_synchronous_search": True, "batch256_last_pipelined_batch_equals_synchronous_search": True}
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # the corpus encode over the same "devices" from the same process: four logical replicas of the layer stack on cuda:0
    er = side["extra"]["corpus_embed_bf16_replicas"]
    assert "error" not in er, er
    assert er["replicas"]["encode_replicas"] == ["cuda:0"] * 4 and er["one_replica"]["encode_replicas"] == ["cuda:0"]
    assert out["config"]["x_corpus_embed_bf16_chunks_per_s"] == pytest.approx(er["replicas"]["value"], rel=1e-4) and er["replicas"]["value"] > 0


@pytest.mark.timeout(1500)
def test_bench_eight_ranks_sharing_one_gpu():
    """The driver's 8-GPU command line in the only form a one-GPU box can run: `bench.py --gpus 8 --backend gloo --share-device`
    starts EIGHT ranks (one process each, all on cuda:0, an eighth of the rows each), runs the timed regions at B = 64 and B = 256
    with the packed candidate exchange, gathers the per-rank rows, and rank 0 then drives the same workload through ONE process with
    eight logical shards.  Asserts the JSON line's multi-rank fields end to end: `per_rank` x 8, `exchange_bindings`,
    `single_process`, every `verified` bit."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-device", "--rows", "80000",
           "--steps", "4", "--warmup", "1", "--repeats", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=700, env=env)
    if r.returncode != 0:        # eight cold interpreters importing torch at once can miss the rendezvous on a fresh box: once more, warm
        first = [l for l in r.stderr.splitlines() if "Gloo" not in l][-15:]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=700, env=env)
        assert r.returncode == 0, ("first attempt:", first, "second attempt:", [l for l in r.stderr.splitlines() if "Gloo" not in l][-25:])
    out, side = _line_and_side(r.stdout)
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["config"]["repeats"] == 2 and out["value"] > 0
    assert out["config"]["value_min"] <= out["value"] <= out["config"]["value_max"] and len(side["ms_per_step_all"]) == 2
    assert all(v is True for v in out["verified"].values()) and len(out["verified"]) == 2, out["verified"]
    eb = side["exchange_bindings"]
    assert eb["backend"] == "gloo" and eb["torch_world_size"] == 8 and eb["share_device"] is True and eb["batch64"] == "torch" and eb["batch256"] == "torch"
    assert [p["rank"] for p in side["per_rank"]] == list(range(8)) and [p["rows"] for p in side["per_rank"]] == [10000] * 8
    assert all(p["batch64"] and p["batch256"] for p in side["per_rank"])
    assert out["roofline"]["rows_per_gpu"] == 10000 and out["config"]["sharding"] == "rows/8"
    sp = side["single_process"]
    assert "error" not in sp, sp
    assert sp["n_gpus"] == 8 and sp["value"] > 0 and "one process, 8 shard(s)" in sp["process_model"]
    assert all(v is True for v in sp["verified"].values()) and len(sp["verified"]) == 2
