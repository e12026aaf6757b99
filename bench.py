#!/usr/bin/env python
"""bench.py — brute-force top-k QPS of the HIP dense-retrieval engine on synthetic 768-d corpora.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Step      = one batch of B queries against the whole (row-sharded) corpus: query packing, the fused
            MFMA scan + top-k kernel, candidate merge; for N>1 also the RCCL all-gather of the
            per-shard [B,k] candidates and the final merge.  Inputs (corpus, queries) are resident
            in HBM before the timed region.
Workload  = north_star's quoted target: 10 M x 768 bf16 rows, B=64, k=20, total corpus FIXED as N
            grows (strong scaling; rank r holds rows [r*10M/N, (r+1)*10M/N)).  BASELINE config 2
            (1 M rows) is measured too at N=1 and reported under "extra".
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event time of
the scan kernel, algorithmic bytes) and `cpu_baseline` (oracle = numpy restatement of the
reference's dense_passage_retrieval, on this box's host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: measured float4 copy


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def gen_rows_dev(torch, lo, hi, dim, device, block=250_000):
    """Seeded standard-normal rows, L2-normalised in fp32, generated per 250k-row block from
    seed (1234, block) so any sharding sees the same global corpus."""
    b0 = lo // block
    for b in range(b0, (hi + block - 1) // block):
        g = torch.Generator(device=device)
        g.manual_seed(1234 * 1_000_003 + b)
        x = torch.randn((block, dim), generator=g, device=device, dtype=torch.float32)
        x = x / x.norm(dim=1, keepdim=True)
        s, e = max(lo, b * block), min(hi, (b + 1) * block)
        yield x[s - b * block:e - b * block].contiguous()


def build_shard(torch, args, rows, rank, world, device):
    from comorag_amd.sharded import ShardedIndex, shard_bounds
    lo, hi = shard_bounds(rows, world, rank)
    sh = ShardedIndex(args.dim, args.dtype, device=device.index, rank=rank, world=world, base=lo, capacity_hint=hi - lo)
    for blk in gen_rows_dev(torch, lo, hi, args.dim, device):
        sh.local.append_dev(blk)
    torch.cuda.synchronize(device)
    return sh


def run_steps(torch, dist, sh, q, k, steps, warmup, world, device):
    for i in range(warmup):
        sh.search_pipelined(q, k, i & 1)
    torch.cuda.synchronize(device)
    sh.local.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        sh.search_pipelined(q, k, i & 1)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    sh.local.profile(False)
    prof = sh.local.profile_collect()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, prof


def cpu_baseline(args, seconds):
    """Oracle (numpy restatement of ComoRAG.dense_passage_retrieval, ComoRAG.py:950-967: np.dot +
    min-max + full argsort, fp32, OpenBLAS threads = all cores) on a 1M-row slice of the workload."""
    from oracle import retrieval_np as orc
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    n = min(args.rows, 1_000_000)
    X = np.concatenate([orc.synthetic_corpus(min(250_000, n - s), args.dim, seed=1234, block=s // 250_000)
                        for s in range(0, n, 250_000)])
    Q = orc.synthetic_queries(args.batch, args.dim, seed=4321)
    orc.dense_passage_retrieval(X, Q[:1])  # warm
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds and done < 4 * args.batch:
        orc.dense_passage_retrieval(X, Q[done % args.batch:done % args.batch + 1])
        done += 1
    dt = time.perf_counter() - t0
    qps_sample = done / dt
    scale = n / args.rows
    return {"value": qps_sample * scale, "unit": "queries/s", "cores": int(blas_threads), "kind": "port",
            "sample": f"{done} single-query dense_passage_retrieval calls (np.dot+min-max+argsort, fp32) over a "
                      f"{n}-row slice in {dt:.1f}s = {qps_sample:.2f} q/s; linearly scaled x{scale:g} to {args.rows} rows",
            "host_cpus": os.cpu_count()}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; comorag_amd has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    from comorag_amd import _lib as L
    info = L.device_info(local_rank)

    g = torch.Generator(device=device)
    g.manual_seed(4321)
    q = torch.randn((args.batch, args.dim), generator=g, device=device, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()

    sh = build_shard(torch, args, args.rows, rank, world, device)
    dt, prof = run_steps(torch, dist, sh, q, args.k, args.steps, args.warmup, world, device)
    qps = args.batch * args.steps / dt
    scan_ms = prof["total_ms"] / max(prof["launches"], 1)
    ach = prof["bytes_per_launch"] / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    out = {
        "metric": "top-k queries/sec", "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"brute-force top-{args.k} over {args.rows} x {args.dim} {args.dtype} rows, batch {args.batch} "
                               f"(north_star target config; corpus fixed, row-sharded over {world} GPU(s))",
                   "rows": args.rows, "dim": args.dim, "batch": args.batch, "k": args.k,
                   "sharding": f"rows/{world}", "device": info["name"], "n_cu": info["n_cu"]},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "frac_of_achievable_6290": ach / HBM_ACHIEVABLE_GBS, "traffic": None,
                     "kernel": "scan_kernel (fused MFMA scan + top-k)", "kernel_ms": scan_ms,
                     "algorithmic_bytes_per_launch": prof["bytes_per_launch"], "launches_timed": prof["launches"],
                     "rows_per_gpu": len(sh)},
    }
    sh.local.close()
    del sh
    if rank == 0 and world == 1 and not args.no_extra and args.rows != 1_000_000:
        # BASELINE config 2: 1 M x 768 bf16, B=64, k=20 on one GPU
        sh2 = build_shard(torch, args, 1_000_000, 0, 1, device)
        dt2, prof2 = run_steps(torch, dist, sh2, q, args.k, max(args.steps, 100), args.warmup, 1, device)
        ms2 = prof2["total_ms"] / max(prof2["launches"], 1)
        out["extra"] = {"config2_1M_rows": {"value": args.batch * max(args.steps, 100) / dt2, "unit": "queries/s",
                                            "ms_per_step": dt2 / max(args.steps, 100) * 1e3, "kernel_ms": ms2,
                                            "hbm_GBps": prof2["bytes_per_launch"] / (ms2 * 1e-3) / 1e9 if ms2 else 0.0}}
        sh2.local.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
