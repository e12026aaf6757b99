"""HBM traffic of the headline scan kernel measured by the run that reports it: bench.py re-runs ITSELF (a few steps, no extras)
under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two do not fit the TCC's counter slots together,
/opt/skills/guides/MI355X_MICROARCH.md) and reads the per-dispatch counters of the main scan from the rocpd database.
gfx950 correction of the same guide: FETCH_SIZE tallies the 128-byte requests of a wide streaming read at 64 bytes -> x 2;
both counters are in KiB."""
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile


def _main_scan_counter(db_path, counter, kernel_substr="scan_kernel"):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]       # noqa: E731
    kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    rows = cur.execute(f"""select s.display_name, d.grid_size_x / d.workgroup_size_x, p.name, e.value, d.end - d.start, d.id
                           from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {ip} p on e.pmc_id = p.id""").fetchall()
    per_dispatch = {}
    for name, grid, ctr, val, dur, did in rows:
        if kernel_substr not in name or ctr != counter:
            continue
        key = (name.split("(")[0], grid)
        per_dispatch.setdefault(key, {}).setdefault(did, [0.0, dur])[0] += val      # a counter may come in several rows per dispatch (per XCC / SE)
    if not per_dispatch:
        return None
    # the main pass = the (kernel, grid) with the largest total duration
    key = max(per_dispatch, key=lambda k_: sum(v[1] for v in per_dispatch[k_].values()))
    vals = [v[0] for v in per_dispatch[key].values()]
    durs = [v[1] / 1e3 for v in per_dispatch[key].values()]
    return {"kernel": key[0], "grid": key[1], "launches": len(vals), "avg_KB": sum(vals) / len(vals), "min_KB": min(vals), "max_KB": max(vals),
            "avg_kernel_us": sum(durs) / len(durs)}


def measure(bench_argv, timeout=200.0):
    """bench_argv: the argument list that reproduces this run's workload (rows / dim / batch / k / dtype ...).
    Returns {"fetch_bytes_corrected_x2", "write_bytes", "traffic_bytes_per_launch", "raw": {...}, "commands": [...]} or {"error": ...}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    raw, cmds = {}, []
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="cmr_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "w", "--", sys.executable, bench, *bench_argv,
                   "--steps", "4", "--warmup", "1", "--no-extra", "--no-cpu-baseline", "--no-pmc"]
            cmds.append(" ".join(["rocprofv3", "--pmc", counter, "--kernel-trace", "--", "python", "bench.py", *bench_argv, "--steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-pmc"]))
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                _, se = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                p.communicate()
                return {"error": f"{counter} pass: timeout after {timeout:.0f} s"}
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return {"error": f"{counter} pass: exit code {p.returncode}, {len(dbs)} database(s)", "stderr_tail": se[-400:]}
            r = _main_scan_counter(dbs[0], counter)
            if r is None:
                return {"error": f"{counter} pass: no scan_kernel dispatch with that counter in the database"}
            raw[counter] = r
        except Exception as e:      # noqa: BLE001  (the headline line must still print)
            return {"error": f"{counter} pass: {e!r}"[:300]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch = raw["FETCH_SIZE"]["avg_KB"] * 1024 * 2
    write = raw["WRITE_SIZE"]["avg_KB"] * 1024
    return {"fetch_bytes_corrected_x2": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write, "raw": raw, "commands": cmds}
