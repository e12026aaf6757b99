"""What bench.py prints, and how it is read back.

The driver keeps the LAST stdout line that starts with `{` and parses it; round 5's line had grown to 22 KB (16 KB of it the
`extra` tree the driver's parser drops) and was not parsed at all.  The contract here:

* the FINAL stdout line is ONE JSON object of at most LINE_LIMIT bytes: the contract keys, `config` (workload + the flat `x_*`
  secondary numbers), `roofline`, a reduced `cpu_baseline`, `verified`;
* everything else (`extra`, `pmc_this_run`, per-rank rows, the single-process leg, the prose notes) is printed on an EARLIER line
  that starts with `EXTRA ` (so the driver's tail still shows it, and no second line starts with `{`) and written to
  `bench_extra.json` beside bench.py.

No torch / numpy imports: the CPU-tier test (tests/test_bench_line.py) builds the line from a canned record.
"""
from __future__ import annotations

import glob
import json
import os
import threading
import time

LINE_LIMIT = 4096
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
LINE_KEYS = CONTRACT_KEYS + ("config", "roofline", "cpu_baseline", "verified")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_source", "kernel", "kernel_ms", "launches_timed",
                 "algorithmic_bytes_per_launch", "rows_per_gpu", "two_scan_streams", "clock")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "batched_value", "batched_cores", "recall_at_k_vs_cpu_fp32", "host_cpus")
# keys of `config` that may be dropped (last first) if a line would still exceed LINE_LIMIT
CONFIG_DROP_ORDER = ("generator", "exchange", "query_batches_rotated", "process_model", "shard_rows")


def _round(v, sig=6):
    """Floats to `sig` significant digits (the line is a record of measurements, not a checkpoint), containers recursively."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        if v.is_integer() and abs(v) < 2 ** 53:       # byte counts and the like stay exact
            return int(v) if abs(v) >= 1e6 else v
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _round(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_round(x, sig) for x in v]
    return v


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def compact(out):
    """(line, side): `line` is the dict of the final stdout line, `side` everything that does not fit the contract."""
    line, side = {}, {}
    for k, v in out.items():
        (line if k in LINE_KEYS else side)[k] = v
    data = str(line.get("data", "synthetic"))
    if data != "synthetic":
        side["data_note"] = data
        line["data"] = "synthetic"
    cfg = dict(line.get("config") or {})
    if "workload" in cfg:
        cfg["workload"] = _clip(cfg["workload"], 118)       # the driver's parser clips strings at ~120 characters
    for k in ("exchange", "generator", "process_model", "device"):
        if k in cfg:
            if isinstance(cfg[k], str) and len(cfg[k]) > 96:
                side.setdefault("config_full", {})[k] = cfg[k]
            cfg[k] = _clip(cfg[k], 96)
    line["config"] = cfg
    roof = line.get("roofline")
    if isinstance(roof, dict):
        side["roofline_full"] = roof
        r = {k: roof[k] for k in ROOFLINE_KEYS if k in roof}
        if "traffic_source" in r:
            r["traffic_source"] = _clip(r["traffic_source"], 60)
        r["kernel"] = _clip(r.get("kernel"), 60)
        line["roofline"] = r
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        side["cpu_baseline_full"] = cpu
        c = {k: cpu[k] for k in CPU_KEYS if k in cpu}
        c["sample"] = _clip(c.get("sample", ""), 118)
        line["cpu_baseline"] = c
    line.setdefault("cpu_baseline", None)
    line.setdefault("verified", {})
    line = _round(line)
    # last resort: never let the line outgrow the driver (drop descriptive config keys, then the x_* flats from the end)
    def size():
        return len(json.dumps(line))
    dropped = []
    for k in CONFIG_DROP_ORDER:
        if size() <= LINE_LIMIT:
            break
        if k in line["config"]:
            dropped.append(k); del line["config"][k]
    xs = [k for k in line["config"] if k.startswith("x_")]
    while size() > LINE_LIMIT and xs:
        k = xs.pop()
        dropped.append(k); del line["config"][k]
    if dropped:
        side["dropped_from_line"] = dropped
    return line, side


def emit(out, stream=None, side_path=None):
    """Print `EXTRA {...}` (when there is anything) and then the final line; write the side file.  Returns the line dict."""
    import sys
    stream = stream or sys.stdout
    line, side = compact(out)
    if side:
        blob = json.dumps(side)
        stream.write("EXTRA " + blob + "\n")
        if side_path:
            try:
                with open(side_path, "w") as f:
                    json.dump({"line": line, "side": side}, f, indent=1)
            except OSError:
                pass
    text = json.dumps(line)
    assert len(text) <= LINE_LIMIT, len(text)
    stream.write(text + "\n")
    stream.flush()
    return line


def parse_emitted(stdout):
    """(line, side) from a bench.py stdout: exactly one line may start with `{`."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    if len(lines) != 1:
        raise ValueError(f"{len(lines)} stdout lines start with '{{' (the contract is exactly one)")
    side = {}
    for l in stdout.splitlines():
        if l.startswith("EXTRA {"):
            side = json.loads(l[6:])
    return json.loads(lines[0]), side


# ------------------------------------------------------------------------------------------------ clocks / power
def _hwmon_dir(pci_bus_id=None):
    """hwmon directory of the amdgpu device (by PCI address when known, else the only / first amdgpu card)."""
    cands = []
    if pci_bus_id:
        cands += glob.glob(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/hwmon/hwmon*")
    if not cands:
        cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    for c in cands:
        if os.path.exists(os.path.join(c, "freq1_input")) or glob.glob(os.path.join(c, "power1_*")):
            return c
    return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class ClockSampler:
    """Shader clock (hwmon freq1_input, Hz) and socket power (power1_average | power1_input, uW) of the GPU, sampled from a helper
    thread while a timed region runs — so that a box-to-box or in-run spread of the kernel time comes with its clock / power
    state.  Reads two sysfs files every `period` seconds; falls back to one amdsmi query before / after when hwmon is unreadable."""

    def __init__(self, pci_bus_id=None, period=0.01):
        self.dir = _hwmon_dir(pci_bus_id)
        self.period = period
        self.f_clk = os.path.join(self.dir, "freq1_input") if self.dir else None
        self.f_pow = None
        if self.dir:
            for n in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(self.dir, n)):
                    self.f_pow = os.path.join(self.dir, n)
                    break
        self.clk, self.pow = [], []
        self._stop = threading.Event()
        self._th = None
        self.idle = self._once()

    def _once(self):
        c = _read_int(self.f_clk) if self.f_clk else None
        p = _read_int(self.f_pow) if self.f_pow else None
        return (c / 1e6 if c else None, p / 1e6 if p else None)

    def _run(self):
        while not self._stop.is_set():
            c, p = self._once()
            if c:
                self.clk.append(c)
            if p:
                self.pow.append(p)
            time.sleep(self.period)

    def start(self):
        if self.dir and self._th is None:
            self._stop.clear()
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def stop(self):
        if self._th is not None:
            self._stop.set()
            self._th.join(1.0)
            self._th = None
        return self.summary()

    def summary(self):
        if not self.dir:
            return {"source": None}
        out = {"source": "hwmon", "samples": len(self.clk), "idle_sclk_mhz": self.idle[0], "idle_power_w": self.idle[1]}
        if self.clk:
            s = sorted(self.clk)
            out.update({"sclk_mhz_median": s[len(s) // 2], "sclk_mhz_min": s[0], "sclk_mhz_max": s[-1]})
        if self.pow:
            out.update({"power_w_mean": sum(self.pow) / len(self.pow), "power_w_max": max(self.pow)})
        return out


def smi_snapshot(device=0):
    """One amdsmi reading (current gfx clock MHz, socket power W, junction temperature) — the before / after record when no hwmon
    sampler exists; {} when the library is not there."""
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        try:
            h = amdsmi.amdsmi_get_processor_handles()[device]
            out = {}
            try:
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                out["sclk_mhz"] = c.get("clk") if isinstance(c, dict) else None
            except Exception:       # noqa: BLE001
                pass
            try:
                p = amdsmi.amdsmi_get_power_info(h)
                out["power_w"] = p.get("current_socket_power") or p.get("average_socket_power")
            except Exception:       # noqa: BLE001
                pass
            return out
        finally:
            amdsmi.amdsmi_shut_down()
    except Exception:       # noqa: BLE001
        return {}
