"""Turn the per-counter summaries of tools/collect_profiles.sh (one JSON line per kernel / grid / counter, written by
tools/rocpd_pmc.py) into the two PMC files under profiles/:
    python tools/summarise_profiles.py gpurun_out/r2_profiles  ->  <dir>/r2_pmc_hbm_traffic.json, <dir>/r2_pmc_wide.json
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count kilobytes; gfx950 tallies 128-B fetch requests at 64 B, hence FETCH x 2."""
import json, os, sys
ROUND = os.environ.get("ROUND", "r5")

d = sys.argv[1]


def lines(name):
    p = os.path.join(d, name)
    return [json.loads(l) for l in open(p) if l.strip().startswith("{")] if os.path.exists(p) else []


def main_pass(rows, counter):
    rows = [r for r in rows if r["counter"] == counter]
    return max(rows, key=lambda r: r["avg_kernel_us"] * r["launches"]) if rows else None


# ---- headline kernel: HBM traffic
f, w = main_pass(lines("pmc_FETCH_SIZE.jsonl"), "FETCH_SIZE"), main_pass(lines("pmc_WRITE_SIZE.jsonl"), "WRITE_SIZE")
if f and w:
    rows, dim, batch, k = 10_000_000, 768, 64, 20
    alg = rows * dim * 2 + batch * dim * 4 + batch * k * 12
    fetch = f["avg"] * 1024 * 2
    write = w["avg"] * 1024
    out = {"kernel": f"{f['kernel']} main pass (grid {f['grid']}, pipelined mode)",
           "workload": {"rows": rows, "dim": dim, "dtype": "bf16", "batch": batch, "k": k},
           "raw": {"FETCH_SIZE": {"launches": f["launches"], "avg_KB": f["avg"], "min_KB": f["min"], "max_KB": f["max"], "avg_kernel_us": f["avg_kernel_us"]},
                   "WRITE_SIZE": {"launches": w["launches"], "avg_KB": w["avg"], "min_KB": w["min"], "max_KB": w["max"], "avg_kernel_us": w["avg_kernel_us"]}},
           "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
           "algorithmic_bytes_per_launch": float(alg), "traffic_over_algorithmic": (fetch + write) / alg,
           "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-pmc",
                        "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-pmc"],
           "collected_with": "tools/collect_profiles.sh", "summarised_with": "tools/rocpd_pmc.py + tools/summarise_profiles.py",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B). Counters are per dispatch: the next "
                   "batch's sampling kernels overlap the main pass but are separate dispatches."}
    json.dump(out, open(os.path.join(d, f"{ROUND}_pmc_hbm_traffic.json"), "w"), indent=1)
    print("traffic / algorithmic:", out["traffic_over_algorithmic"])

# ---- wide kernel: SQ counters (values are per shader engine: rocpd stores 32 rows per dispatch, the tool averages them)
raw = {}
kernel_us, grid = None, None
for name in ("pmc_wide_a.jsonl", "pmc_wide_b.jsonl"):
    rows = lines(name)
    if not rows:
        continue
    big = max(rows, key=lambda r: r["avg_kernel_us"])           # the main pass (the sampling passes are the short ones)
    for r in rows:
        if r["grid"] == big["grid"]:
            raw[r["counter"]] = r["avg"]
            if name == "pmc_wide_a.jsonl":
                kernel_us, grid = r["avg_kernel_us"], r["grid"]
fw, ww = main_pass(lines("pmc_wide_FETCH_SIZE.jsonl"), "FETCH_SIZE"), main_pass(lines("pmc_wide_WRITE_SIZE.jsonl"), "WRITE_SIZE")
if raw and kernel_us:
    n_se = 32
    mfma = raw.get("SQ_INSTS_MFMA", 0.0)
    busy = raw.get("SQ_BUSY_CYCLES", 0.0)
    wave_cycles = raw.get("SQ_WAVE_CYCLES", 0.0)
    # per shader engine: SQ_BUSY_CYCLES = shader clocks the engine was busy (= the kernel's duration in clocks); the matrix
    # pipes of its active SIMDs (workgroups / 32 engines x 4, one workgroup per CU) can be busy that long each
    clock_ghz = busy / (kernel_us * 1e3) if busy else None
    simds_per_se = int(grid.split("x")[0]) / n_se * 4
    flop = 2.0 * 256 * 10_000_000 * 768
    derived = {"shader_clock_GHz": clock_ghz,
               "mfma_pipe_busy_frac": raw.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (busy * simds_per_se) if busy else None,
               "achieved_TFLOPs": flop / (kernel_us * 1e-6) / 1e12,
               "wave_time_split": {"parked_at_waitcnt_or_barrier (SQ_WAIT_ANY)": raw.get("SQ_WAIT_ANY", 0.0) / wave_cycles if wave_cycles else None,
                                   "issue_stalled (SQ_WAIT_INST_ANY)": raw.get("SQ_WAIT_INST_ANY", 0.0) / wave_cycles if wave_cycles else None,
                                   "issuing (SQ_ACTIVE_INST_ANY)": raw.get("SQ_ACTIVE_INST_ANY", 0.0) / wave_cycles if wave_cycles else None},
               "valu_per_mfma": raw.get("SQ_INSTS_VALU", 0.0) / mfma if mfma else None,
               # SQ_INSTS_VALU counts the MFMAs themselves (VALU-class instructions): round 4's 1.95 was 1.00 MFMA + 0.95 others, which is
               # what the kernel's listing holds (96 MFMAs + 68 VALU on the hot path of a panel + the slow path)
               "valu_non_mfma_per_mfma": raw.get("SQ_INSTS_VALU", 0.0) / mfma - 1.0 if mfma else None,
               "non_mfma_instructions_per_mfma": (raw.get("SQ_INSTS_VALU", 0.0) - mfma + raw.get("SQ_INSTS_SALU", 0.0) + raw.get("SQ_INSTS_LDS", 0.0)
                                                  + raw.get("SQ_INSTS_VMEM", 0.0)) / mfma if mfma else None,
               "salu_per_mfma": raw.get("SQ_INSTS_SALU", 0.0) / mfma if mfma else None,
               "lds_instr_per_mfma": raw.get("SQ_INSTS_LDS", 0.0) / mfma if mfma else None,
               "vmem_instr_per_mfma": raw.get("SQ_INSTS_VMEM", 0.0) / mfma if mfma else None,
               "lds_bank_conflict_cycles": raw.get("SQ_LDS_BANK_CONFLICT")}
    if fw and ww:
        t = fw["avg"] * 1024 * 2 + ww["avg"] * 1024
        derived["hbm_traffic_bytes_per_launch"] = t
        derived["traffic_over_algorithmic"] = t / (10_000_000 * 768 * 2 + 256 * 768 * 4 + 256 * 20 * 12)
    out = {"kernel": "scan_wide_kernel<bf16,KS=48,NT=2,CAP=128,NST=8> main pass, 10 M x 768 bf16 rows, batch 256, pipelined mode (one CU per shader "
                     "engine reserved for the next batch's sampling), software-pipelined epilogue",
           "kernel_ms_under_the_profiler": kernel_us / 1e3, "units": f"counter values are per shader engine (the rocpd rows: {n_se} per dispatch), averaged over the dispatches",
           "raw_per_shader_engine": raw, "derived": derived,
           "collected_with": "tools/collect_profiles.sh", "summarised_with": "tools/rocpd_pmc.py + tools/summarise_profiles.py"}
    json.dump(out, open(os.path.join(d, f"{ROUND}_pmc_wide.json"), "w"), indent=1)
    print("wide: valu/mfma", derived["valu_per_mfma"], "mfma busy", derived["mfma_pipe_busy_frac"])
