"""The line bench.py hands the driver (CPU tier).  Round 5's 22 KB line was not parsed by the driver at all; tools/bench_line.py now
splits a bench record into a final line of <= 4 KB (contract keys, `config` with the flat x_* numbers, `roofline`, reduced
`cpu_baseline`, `verified`) and a side record printed earlier as `EXTRA {...}`.  The canned record is round 5's own output
(profiles/r5_bench_n1.json: every `extra` row, the PMC passes, the prose notes)."""
import io
import json
import os

import pytest

from tools import bench_line as bl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline", "verified"}


def _canned():
    with open(os.path.join(ROOT, "profiles", "r5_bench_n1.json")) as f:
        txt = f.read()
    rec = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
    assert len(json.dumps(rec)) > 20_000 and "extra" in rec            # the record that broke the driver's parser
    return rec


def test_final_line_is_small_and_carries_the_contract():
    rec = _canned()
    line, side = bl.compact(rec)
    text = json.dumps(line)
    assert len(text) < bl.LINE_LIMIT == 4096, len(text)
    assert set(line) == CONTRACT
    assert line["data"] == "synthetic" and "drawn by" in side["data_note"]
    assert line["metric"] == "top-k queries/sec" and line["value"] == pytest.approx(rec["value"], rel=1e-5)
    assert line["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-5)
    # roofline: the task's keys, numerically the record's
    r = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r)
    assert r["frac"] == pytest.approx(rec["roofline"]["frac"], rel=1e-5) and r["traffic"] == pytest.approx(rec["roofline"]["traffic"], rel=1e-5)
    assert r["achieved"] / r["peak"] == pytest.approx(r["frac"], rel=1e-4)
    # cpu_baseline: reduced to the stated keys, `sample` kept (clipped)
    c = line["cpu_baseline"]
    assert set(c) <= set(bl.CPU_KEYS) and {"value", "unit", "cores", "kind", "sample"} <= set(c) and len(c["sample"]) <= 118
    assert c["batched_value"] == pytest.approx(rec["cpu_baseline"]["batched_value"], rel=1e-5)
    # the flat secondary numbers survive where the driver's parser keeps them
    xs = {k for k in line["config"] if k.startswith("x_")}
    assert {"x_config3_batch256_kernel_ms", "x_config4_call_frac_of_hbm", "x_single_query_latency_1M_rows_us", "x_corpus_embed_bf16_chunks_per_s"} <= xs
    assert all(len(v) <= 120 for v in line["config"].values() if isinstance(v, str))
    assert line["verified"] == rec["verified"]
    # nothing is lost: what left the line is in the side record
    assert side["extra"] == rec["extra"] and side["roofline_full"] == rec["roofline"] and side["cpu_baseline_full"] == rec["cpu_baseline"]
    assert "dropped_from_line" not in side


def test_emit_prints_exactly_one_brace_line_and_an_extra_line(tmp_path):
    rec = _canned()
    buf = io.StringIO()
    side_file = tmp_path / "bench_extra.json"
    line = bl.emit(rec, stream=buf, side_path=str(side_file))
    out = buf.getvalue().splitlines()
    assert [l[:1] for l in out].count("{") == 1 and out[-1].startswith("{") and out[0].startswith("EXTRA {")
    assert json.loads(out[-1]) == line and len(out[-1]) < 4096
    got_line, got_side = bl.parse_emitted(buf.getvalue())
    assert got_line == line and got_side["extra"].keys() == rec["extra"].keys()
    saved = json.loads(side_file.read_text())
    assert saved["line"] == line and saved["side"]["extra"].keys() == rec["extra"].keys()


def test_an_oversized_config_sheds_keys_instead_of_growing_the_line():
    rec = _canned()
    rec["config"].update({f"x_filler_{i:03d}": "y" * 100 for i in range(60)})       # 7 KB of extra flats
    line, side = bl.compact(rec)
    assert len(json.dumps(line)) <= bl.LINE_LIMIT
    assert side["dropped_from_line"] and all(k.startswith("x_filler_") or k in bl.CONFIG_DROP_ORDER for k in side["dropped_from_line"])
    assert "workload" in line["config"] and "x_config3_batch256_kernel_ms" in line["config"]
    with pytest.raises(ValueError):
        bl.parse_emitted('{"a": 1}\n{"b": 2}\n')


def test_bench_py_prints_through_emit_only():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(" not in src and src.count("emit(out") == 3


def test_clock_sampler_without_a_gpu_is_inert():
    s = bl.ClockSampler("0000:ff:1f.0")
    if s.dir is None:                       # no amdgpu hwmon in this container
        assert s.start().stop() == {"source": None}
    else:
        s.start(); summary = s.stop()
        assert summary["source"] == "hwmon"
