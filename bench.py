#!/usr/bin/env python
"""bench.py — brute-force top-k QPS of the HIP dense-retrieval engine on synthetic 768-d corpora.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Launched either by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(WORLD_SIZE / RANK / LOCAL_RANK in the environment) or plainly as `python bench.py --gpus N`, in which case this script
re-executes itself under torch.distributed.run on a free port (it fails with a clear message when the node shows fewer
than N GPUs).  `--backend gloo --share-device` runs the whole N > 1 control flow with all ranks on cuda:0 and the
candidate exchange staged through the host — the rehearsal a 1-GPU box allows (tests/test_bench_multirank_gpu.py).

Step      = one batch of B queries against the whole (row-sharded) corpus: query packing, sampling passes, the fused
            MFMA scan + top-k kernel, candidate merge; for N>1 also the ONE RCCL all-gather of the packed per-shard
            [B,k] candidates and the final merge.  Inputs (corpus, queries) are resident in HBM before the timed region.
Workload  = north_star's quoted target: 10 M x 768 bf16 rows, B=64, k=20, total corpus FIXED as N grows (strong
            scaling; rank r holds rows [r*10M/N, (r+1)*10M/N)).  BASELINE config 3 as written (the same corpus, batch
            256) is timed too at every N ("config3_batch256"); BASELINE config 2 (1 M rows) and the other extras at N=1.
`--single-process --gpus N` times the SAME workload on the index the drop-in API builds for `global_config.num_shards = N`: one
process, N devices (`comorag_amd.multi_index.MultiDeviceIndex` — per-shard worker threads enqueue, every shard's merge kernel
writes its candidates into mapped host memory, host-side final merge); under a launcher with N > 1 ranks, rank 0 runs that leg
in a child process after the one-rank-per-GPU measurement and reports it as `single_process` in the same JSON line.
After the timed loop the LAST pipelined batch is compared with a synchronous search of the same batch (must be
bit-identical), and recall@k against the fp32 CPU ranking is computed from THOSE ids.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event time of the scan kernel on
its own stream, algorithmic bytes, shader clock / power sampled during the timed regions) and `cpu_baseline` (the oracle's
restatement of the reference's CPU code on this box's host cores, bounded samples).  That final line is <= 4 KB
(tools/bench_line.py: contract keys, `config` with the flat `x_*` secondary numbers, `roofline`, reduced `cpu_baseline`,
`verified`); the full tree (`extra`, PMC passes, per-rank rows, prose notes) goes to an earlier stdout line that starts with
`EXTRA ` and to `bench_extra.json` beside this script.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_line import ClockSampler, emit, parse_emitted  # noqa: E402  (no torch / numpy inside)

SIDE_FILE = os.path.join(ROOT, "bench_extra.json")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
# HIP events bracket every PROFILE_EVERY-th main scan of the timed region: the two event packets cost ~25 us on the scan stream
# (9 % of a 1 M-row step), so timing every launch would slow what `value` reports; tools/pipe_only.py times none at all
PROFILE_EVERY = 4
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: measured float4 copy
MFMA_BF16_PEAK_TFLOPS = 2500.0   # same guide: dense bf16 peak (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="the timed K-step region is run this many times back to back (same warm state, each one bracketed by "
                                                            "barrier + synchronize); value / ms_per_step are the MEDIAN region, min / max beside them")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--exchange", default="torch", choices=["torch", "cabi"],
                    help="N>1: binding of the batch-64 exchange — torch.distributed collective or the library's own RCCL call; batch 256 runs through the OTHER one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="N>1: nccl (= RCCL over xGMI) or gloo (host collective; with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="N>1: every rank on cuda:0 (1-GPU rehearsal of the multi-rank flow; needs --backend gloo)")
    ap.add_argument("--query-batches", type=int, default=4, help="distinct query batches rotated through the timed steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--index-option", action="append", default=[], metavar="NAME=VALUE",
                    help="route selector passed to cmr_index_set_option on this rank's index (A/B runs; DESIGN.md appendix)")
    ap.add_argument("--only-config3", action="store_true", help="after the headline, run the batch-256 row and skip the other extras")
    ap.add_argument("--survey-rng", action="store_true", help="draw the corpus with numpy default_rng([1234, block]) on the host, literally as SURVEY 8(d) words it (slow: minutes at 10 M rows); default: the same distribution from the device generator")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1: do not re-run a few steps under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for roofline.traffic")
    ap.add_argument("--single-process", action="store_true", help="ONE process drives all --gpus devices through MultiDeviceIndex (what hooks.install builds for num_shards = N); no launcher, no collective")
    ap.add_argument("--no-single-process-leg", action="store_true", help="N > 1 under a launcher: skip the single-process leg rank 0 runs afterwards")
    ap.add_argument("--single-process-timeout", type=float, default=300.0)
    ap.add_argument("--extras-timeout", type=float, default=240.0, help="N > 1: seconds the batch-256 / per-rank section may take before the headline line is printed without it")
    return ap.parse_args()


RNG_BLOCK = 1_000_000        # "for N = 10 M generate in 1 M-row blocks with default_rng([1234, blk])"
SURVEY_RNG = False           # --survey-rng: draw the corpus with numpy on the host exactly as SURVEY 8(d) words it (minutes at 10 M rows)


def _host_block(blk, n_rows, dim):
    """Rows [0, n_rows) of block `blk` of the SURVEY 8(d) corpus: rng = default_rng([1234, blk]); X = rng.standard_normal((1 M, D),
    float32), row-L2-normalised in fp32 (a prefix of the block's rows is the prefix of its stream: nothing beyond n_rows is drawn)."""
    x = np.random.default_rng([1234, int(blk)]).standard_normal((int(n_rows), int(dim)), dtype=np.float32)
    x /= np.sqrt(np.einsum("ij,ij->i", x, x, dtype=np.float32))[:, None]
    return x


def gen_rows_dev(torch, lo, hi, dim, device, block=250_000):
    """Rows [lo, hi) of the synthetic corpus as CUDA tensors of <= `block` rows: seeded standard-normal rows, L2-normalised in
    fp32, any sharding sees the same global corpus.  Default: drawn ON THE DEVICE per 250 K-row block from seed (1234, block) —
    the same distribution as SURVEY.md 8(d)'s numpy recipe at a hundredth of its time (numpy draws 65 M samples/s per host thread:
    two minutes of CPU for 10 M x 768, again for every child run); `--survey-rng` draws with numpy `default_rng([1234, blk])`
    per 1 M-row block on host threads instead, literally as 8(d) words it."""
    if not SURVEY_RNG:
        b0 = lo // block
        for b in range(b0, (hi + block - 1) // block):
            g = torch.Generator(device=device)
            g.manual_seed(1234 * 1_000_003 + b)
            x = torch.randn((block, dim), generator=g, device=device, dtype=torch.float32)
            x = x / x.norm(dim=1, keepdim=True)
            s, e = max(lo, b * block), min(hi, (b + 1) * block)
            yield x[s - b * block:e - b * block].contiguous()
        return
    from concurrent.futures import ThreadPoolExecutor
    if hi <= lo:
        return
    blks = list(range(lo // RNG_BLOCK, (hi - 1) // RNG_BLOCK + 1))
    need = {b: min(hi, (b + 1) * RNG_BLOCK) - b * RNG_BLOCK for b in blks}      # rows of block b to draw (from its start)
    workers = max(1, min(len(blks), 4, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(workers) as ex:
        futs = {b: ex.submit(_host_block, b, need[b], dim) for b in blks[:workers + 1]}
        for i, b in enumerate(blks):
            x = futs.pop(b).result()
            nxt = i + workers + 1
            if nxt < len(blks):
                futs[blks[nxt]] = ex.submit(_host_block, blks[nxt], need[blks[nxt]], dim)
            s0 = max(lo, b * RNG_BLOCK) - b * RNG_BLOCK
            for r0 in range(s0, need[b], block):
                yield torch.from_numpy(x[r0:min(r0 + block, need[b])]).to(device)
            del x


def _generator_note():
    """Which generator drew the corpus (goes into the JSON line's `data`)."""
    return ("numpy default_rng([1234, block]) on the host, SURVEY 8(d) literally" if SURVEY_RNG else
            "torch device generator (torch.randn, seed 1234 * 1000003 + block per 250 K-row block), rows L2-normalised in fp32: "
            "SURVEY 8(d)'s distribution, not its numpy bit stream (--survey-rng draws that one)")


def build_shard(torch, args, rows, rank, world, device, host=None, timing=False):
    """host: a preallocated fp32 array [hi - lo, dim] that receives the shard's rows (for the CPU legs)."""
    from comorag_amd.sharded import ShardedIndex, shard_bounds
    lo, hi = shard_bounds(rows, world, rank)
    sh = ShardedIndex(args.dim, args.dtype, device=device.index, rank=rank, world=world, base=lo, capacity_hint=hi - lo,
                      exchange=args.exchange, timing=timing)
    at = 0
    for blk in gen_rows_dev(torch, lo, hi, args.dim, device):
        sh.local.append_dev(blk)
        if host is not None:
            host[at:at + len(blk)] = blk.cpu().numpy()
        at += len(blk)
    torch.cuda.synchronize(device)
    for opt in args.index_option:
        name, _, value = opt.partition("=")
        sh.local.set_option(name.strip(), int(value))
    return sh


def make_queries(torch, n_batches, batch, dim, device, seed):
    """n_batches distinct batches of unit queries (standard-normal, normalised; 10 % of them planted near a
    corpus row, q = normalise(X[j] + 0.1 g) with X[j] among the corpus' first 4096 rows): the timed steps rotate through them,
    so the data-dependent part of a step (sampling thresholds, slow-path frequency) is not one sample.  Every rank draws the same
    batches."""
    out = []
    n_plant = max(1, batch // 10)
    x0 = next(iter(gen_rows_dev(torch, 0, 4096, dim, device))).cpu().numpy()      # the corpus' first rows, whichever generator draws it
    for j in range(n_batches):
        rng = np.random.default_rng([1234, int(seed), j])
        q = rng.standard_normal((batch, dim), dtype=np.float32)
        rows = rng.integers(0, len(x0), n_plant)
        q[:n_plant] = x0[rows] + 0.1 * q[:n_plant] / np.sqrt(dim)       # g scaled to the rows' element size: a near neighbour, not the row itself
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        out.append(torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(device))
    torch.cuda.synchronize(device)
    return out


def _pci_address(torch, device):
    """sysfs PCI address of a torch device ("0000:75:00.0") or None."""
    try:
        p = torch.cuda.get_device_properties(device)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:       # noqa: BLE001
        return None


def run_steps(torch, dist, sh, qs, k, steps, warmup, world, device, every=PROFILE_EVERY, ctl="cuda", repeats=1, sampler=None):
    """qs: list of query batches, step i takes qs[i % len(qs)].  The timed region (exactly `steps` steps between barrier +
    synchronize on both sides, max over ranks) is run `repeats` times back to back after ONE warm-up.
    Returns (median seconds, profile, last batch's buffers, index of the query batch of the last step, all regions' seconds)."""
    for i in range(warmup):
        sh.search_pipelined(qs[i % len(qs)], k, i & 1)["done"].synchronize()
    torch.cuda.synchronize(device)
    sh.times = []
    sh.local.profile(every)       # HIP events around every `every`-th main scan of the timed regions
    dts = []
    last = None
    if sampler is not None:        # shader clock / power while the timed regions run (a helper thread reading two sysfs files every 10 ms)
        sampler.start()
    for _ in range(max(1, repeats)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            last = sh.search_pipelined(qs[i % len(qs)], k, i & 1)
        last["done"].synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if ctl == "cuda" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        dts.append(dt)
    if sampler is not None:
        sampler.stop()
    sh.local.profile(False)
    prof = sh.local.profile_collect()
    ex = sh.exchange_times_ms() if sh.timing else []
    prof["exchange_ms"] = float(np.mean([a for a, _ in ex])) if ex else 0.0
    prof["merge_ms"] = float(np.mean([m for _, m in ex])) if ex else 0.0
    return float(np.median(dts)), prof, last, (steps - 1) % len(qs), dts


def verify_last_batch(sh, last, qh, k):
    """The outputs of the LAST timed pipelined batch vs a synchronous search of the same batch (host-buffer API; for N>1
    the host path with its own all-gather + host merge): must be bit-identical."""
    ids = last["o_ids"].cpu().numpy().copy()
    sc = last["o_sc"].cpu().numpy().copy()
    sid, ssc = sh.search(qh, k)
    ok = bool(np.array_equal(ids, sid) and np.array_equal(sc, ssc))
    return ids, sc, ok


def thread_info():
    info = {"os_cpu_count": os.cpu_count()}
    try:
        from threadpoolctl import threadpool_info
        info["threadpool_info"] = [{k: p.get(k) for k in ("user_api", "internal_api", "num_threads", "version")} for p in threadpool_info()]
        info["blas_threads"] = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        info["blas_threads"] = os.cpu_count() or 1
    try:
        import torch
        info["torch_num_threads"] = torch.get_num_threads()
    except Exception:
        pass
    return info


def cpu_baseline(args, seconds, X, Q, gpu_ids, full_size):
    """The oracle (= numpy / torch-CPU restatement of the reference's code; the reference tree itself is not on the GPU
    box, hence kind "port") on this box's host cores, over the SAME rows the GPU scanned (copied back from HBM):
      (i)   single-query ComoRAG.dense_passage_retrieval (ComoRAG.py:950-967: np.dot + min-max + full argsort, fp32)
      (ii)  batched retrieve_knn forced onto the CPU (utils/embed_utils.py:8-97: torch.mm + torch.topk), B=64, k=20, 1 M keys
      (iii) batch_encode chunks/s on the CPU (BGEEmbedding.py:131-185 on the same random-init BERT-base)
    plus recall@k of the bf16 GPU ids of the timed configuration vs the fp32 CPU ranking of the same rows."""
    from oracle import retrieval_np as orc
    ti = thread_info()
    n = len(X)
    orc.dense_passage_retrieval(X, Q[:1])  # warm
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds and done < 4 * len(Q):
        orc.dense_passage_retrieval(X, Q[done % len(Q):done % len(Q) + 1])
        done += 1
    dt = time.perf_counter() - t0
    qps_sample = done / dt
    scale = n / args.rows
    # recall: fp32 ranking of the same rows, chunked so the [B, n] score block stays small
    import torch
    qt = torch.from_numpy(Q)
    best_s = torch.full((len(Q), args.k), -np.inf); best_i = torch.zeros((len(Q), args.k), dtype=torch.int64)
    step = 1_000_000
    for lo in range(0, n, step):
        s = qt @ torch.from_numpy(X[lo:lo + step]).T
        ts, ti_ = torch.topk(s, min(args.k, s.shape[1]), dim=1)
        cs, ci = torch.cat([best_s, ts], 1), torch.cat([best_i, ti_ + lo], 1)
        o = torch.topk(cs, args.k, dim=1).indices
        best_s, best_i = torch.gather(cs, 1, o), torch.gather(ci, 1, o)
    ref_ids = best_i.numpy()
    recall = float(np.mean([len(set(gpu_ids[i].tolist()) & set(ref_ids[i].tolist())) / args.k for i in range(len(Q))]))
    out = {"value": qps_sample * scale, "unit": "queries/s", "cores": int(ti.get("blas_threads", 1)), "kind": "port",
           "value_is": "SINGLE-query dense_passage_retrieval (what ComoRAG issues: np.dot + min-max + FULL argsort of all rows per query); the like-for-like "
                       "batched top-k comparator is `batched_value` below",
           "kind_note": "oracle/retrieval_np.py, the line-by-line numpy restatement of the reference functions (pinned to reference "
                        "outputs in tests/); the reference tree itself does not exist on the GPU box",
           "sample": f"{done} single-query dense_passage_retrieval calls, {'ALL' if full_size else 'first'} {n} fp32 rows, {dt:.1f}s"
                     + ("" if full_size else f", scaled x{scale:g}"),
           "sample_full": f"{done} single-query dense_passage_retrieval calls (np.dot + min-max + full argsort, fp32) over "
                          f"{'ALL' if full_size else 'the first'} {n} rows of the bench corpus in {dt:.1f}s = {qps_sample:.3f} q/s"
                          + ("" if full_size else f"; linearly scaled x{scale:g} to {args.rows} rows (host RAM too small for the fp32 copy)"),
           "host_cpus": os.cpu_count(), "threads": ti,
           "recall_at_k_vs_cpu_fp32": recall,
           "recall_note": f"top-{args.k} ids of the {args.dtype} HIP index from the LAST TIMED PIPELINED batch of the headline configuration "
                          f"vs the fp32 CPU ranking (torch.mm + topk), {len(Q)} queries, {n} rows"}
    # (ii) batched CPU comparator — the like-for-like CPU leg of `value` (same batch, same k, the SAME rows: utils/embed_utils.py:8-97's
    # torch.mm + torch.topk per 10000-key block, forced onto the CPU); its figure sits at the TOP LEVEL next to the single-query one
    try:
        import torch
        orc.retrieve_knn_torch_cpu(Q[:8], X[:100_000], k=args.k)
        # a BOUNDED sample: the first 2 M rows (the function is a loop over 10000-key blocks: linear in the keys — all 10 M rows took 57 s
        # per batch on 128 threads, gpurun_out/r5e), its time scaled to the full corpus
        nk = min(n, 2_000_000)
        t0 = time.perf_counter(); reps = 0
        while reps < 3 and time.perf_counter() - t0 < max(6.0, seconds / 2):
            orc.retrieve_knn_torch_cpu(Q, X[:nk], k=args.k); reps += 1
        dtb = (time.perf_counter() - t0) / max(reps, 1)
        bt = int(torch.get_num_threads())
        full_b, scale_b = (full_size and nk == n), args.rows / nk
        out["batched_value"] = len(Q) / dtb / scale_b
        out["batched_unit"] = "queries/s"
        out["batched_cores"] = bt
        out["batched_sample"] = (f"{reps} call(s) of retrieve_knn forced onto the CPU (utils/embed_utils.py:8-97: fp32 normalise, torch.mm + torch.topk per 10000-key "
                                 f"block, final topk), batch {len(Q)}, k = {args.k}, over {'ALL' if full_b else 'the first'} {nk} rows of the bench corpus, "
                                 f"{dtb * 1e3:.0f} ms per batch on {bt} torch intra-op threads of {os.cpu_count()} host CPUs"
                                 + ("" if full_b else f"; time linearly scaled x{scale_b:g} to {args.rows} rows"))
        out["batched_retrieve_knn"] = {"value": len(Q) / dtb / scale_b, "unit": "queries/s", "batch": len(Q), "k": args.k, "keys": nk, "ms_per_batch": dtb * 1e3, "threads": bt,
                                       "what": "utils/embed_utils.py:8-97 forced onto the CPU (fp32 normalise, torch.mm + torch.topk per 10000-key block, merge)"}
    except Exception as e:
        out["batched_retrieve_knn"] = {"error": repr(e)[:300]}
    # (iii) CPU encode
    try:
        from oracle import encode_torch as enc
        from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
        tok, words = synthetic_wordpiece_tokenizer()
        model = random_bert("base", vocab_size=len(tok))
        chunks = synthetic_chunks(words, 8)
        enc.encode(model, tok, chunks[:1], instruction=enc.BGE_PREFIX)
        t0 = time.perf_counter()
        enc.batch_encode(model, tok, chunks, batch_size=8) if hasattr(enc, "batch_encode") else enc.encode(model, tok, chunks, instruction=enc.BGE_PREFIX)
        dte = time.perf_counter() - t0
        out["batch_encode"] = {"value": len(chunks) / dte, "unit": "chunks/s", "chunks": len(chunks), "model": "BERT-base shape, random init, fp32, CPU",
                               "what": "embedding_model/BGEEmbedding.py:92-185 restated (tokenise + forward + mean-pool + L2-normalise), ~480-token chunks"}
    except Exception as e:
        out["batch_encode"] = {"error": repr(e)[:300]}
    return out


def host_api_rate(torch, sh, Qh, k, steps=30):
    """Same batch through the host-buffer API (H2D queries, D2H results, one sync per call)."""
    for _ in range(3):
        sh.local.search(Qh, k)
    t0 = time.perf_counter()
    for _ in range(steps):
        sh.local.search(Qh, k)
    dt = time.perf_counter() - t0
    return {"value": len(Qh) * steps / dt, "unit": "queries/s", "ms_per_call": dt / steps * 1e3}


def single_query_latency(torch, args, device, sizes=(6, 1000, 10_000, 100_000)):
    """What ComoRAG's per-question threads issue (ComoRAG.py:937-967 is one query per call): median wall time of a
    synchronous single-query top-k through the host-buffer API, Python wrapper included, per corpus size."""
    from comorag_amd.index import DenseIndex
    out, all_scores = {}, {}
    rng = np.random.default_rng(11)
    for rows in sizes:
        idx = DenseIndex(args.dim, args.dtype, device=device.index or 0, capacity_hint=rows)
        for blk in gen_rows_dev(torch, 0, rows, args.dim, device):
            idx.append_dev(blk)
        torch.cuda.synchronize(device)
        q1 = rng.standard_normal((1, args.dim)).astype(np.float32)
        q1 /= np.linalg.norm(q1)
        kk = min(args.k, rows)
        for _ in range(5):
            idx.search(q1, kk)
        t = []
        for _ in range(50):
            t0 = time.perf_counter()
            idx.search(q1, kk)
            t.append(time.perf_counter() - t0)
        out[str(rows)] = float(np.median(t) * 1e6)
        for _ in range(5):
            idx.scores(q1)
        t = []
        for _ in range(30):
            t0 = time.perf_counter()
            idx.scores(q1)
            t.append(time.perf_counter() - t0)
        all_scores[str(rows)] = float(np.median(t) * 1e6)
        idx.close()
    return {"unit": "us per call (median of 50)", "rows": out,
            "all_scores_rows": all_scores, "all_scores_note": "cmr_index_scores, one query: what dense_passage_retrieval / get_fact_scores call (median of 30)"}


def summarise(batch, steps, dt, prof, rows_gpu, dim, dual=False, dts=None):
    """dual: the pass alternated between the index's two scan streams (short scans, cmr_index_get_option
    "pipe_dual_scan_active"): consecutive launches overlap — a launch's begin-to-end then includes the wait for the CUs of the
    previous scan, it is no duration — so that row's HBM figures are algorithmic bytes / STEP time (a lower bound of what the
    kernel itself achieves) and `kernel_ms` is given as the lifetime it is."""
    ms = prof["total_ms"] / max(prof["launches"], 1)
    step_ms = dt / steps * 1e3
    by = prof["bytes_per_launch"]
    out = {"value": batch * steps / dt, "unit": "queries/s", "batch": batch, "ms_per_step": step_ms,
           "hbm_GBps_step": by / (step_ms * 1e-3) / 1e9 if step_ms else 0.0, "frac_step_of_8TBps": by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if step_ms else 0.0,
           "two_scan_streams": bool(dual), "exchange_ms": prof["exchange_ms"], "merge_ms": prof["merge_ms"]}
    if dts and len(dts) > 1:        # the timed region was repeated: dt is the median region
        out.update({"repeats": len(dts), "ms_per_step_all": [d / steps * 1e3 for d in dts], "ms_per_step_min": min(dts) / steps * 1e3,
                    "ms_per_step_max": max(dts) / steps * 1e3, "value_min": batch * steps / max(dts), "value_max": batch * steps / min(dts)})
    if dual:
        out.update({"kernel_ms": None, "kernel_lifetime_ms": ms, "hbm_GBps": out["hbm_GBps_step"], "frac_of_8TBps": out["frac_step_of_8TBps"],
                    "mfma_TFLOPs": 2.0 * batch * rows_gpu * dim / (step_ms * 1e-3) / 1e12 if step_ms else 0.0,
                    "note": "two alternating scan streams: launches overlap, bytes / step time"})
    else:
        out.update({"kernel_ms": ms, "hbm_GBps": by / (ms * 1e-3) / 1e9 if ms else 0.0, "frac_of_8TBps": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else 0.0,
                    "mfma_TFLOPs": 2.0 * batch * rows_gpu * dim / (ms * 1e-3) / 1e12 if ms else 0.0})
    return out


def single_process_main(args):
    """`--single-process`: the whole node from ONE process — the index hooks.install / EmbeddingStore.device_index build for
    global_config.num_shards = N.  Same workload, same JSON line; `value` = queries / s of pipelined batches whose merged
    results were collected on the host."""
    import torch
    from comorag_amd import _lib as L
    from comorag_amd.multi_index import MultiDeviceIndex
    n = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; comorag_amd has no CPU fallback")
    have = torch.cuda.device_count()
    if not args.share_device and have < n:
        raise SystemExit(f"bench.py --single-process --gpus {n}: this node shows {have} GPU(s); use --share-device for {n} logical shards on cuda:0")
    devices = [0] * n if args.share_device else list(range(n))
    per = (args.rows + n - 1) // n
    mi = MultiDeviceIndex(args.dim, args.dtype, devices=devices, capacity_hint=args.rows, options={"append_block_rows": per})
    blk = 250_000
    for b0, x in zip(range(0, args.rows, blk), gen_rows_dev_multi(torch, args, devices, per, blk)):
        mi.append_dev(x)
    for d in set(devices):
        torch.cuda.synchronize(torch.device("cuda", d))
    for opt in args.index_option:
        name, _, value = opt.partition("=")
        mi.set_option(name.strip(), int(value))
    info = L.device_info(devices[0])
    dev0 = torch.device("cuda", devices[0])

    def run(batch, steps, warmup, seed):
        qs = make_queries(torch, max(1, args.query_batches), batch, args.dim, dev0, seed)
        placed = [mi.place_queries(q) for q in qs]
        inflight = []
        for i in range(warmup):
            mi.collect(mi.search_pipelined(placed[i % len(placed)], args.k))
        shards = [mi.shard(s) for s in range(n)]
        for sh in shards:
            sh.profile(PROFILE_EVERY)
        mi.host_profile(reset=True)
        for d in set(devices):
            torch.cuda.synchronize(torch.device("cuda", d))
        t_collect = t_enq = 0.0
        last = None
        t0 = time.perf_counter()
        for i in range(steps):
            te = time.perf_counter()
            inflight.append(mi.search_pipelined(placed[i % len(placed)], args.k))
            t_enq += time.perf_counter() - te
            if len(inflight) > 2:                      # two batches stay in flight behind the one being merged
                tc = time.perf_counter()
                last = mi.collect(inflight.pop(0))
                t_collect += time.perf_counter() - tc
        while inflight:
            last = mi.collect(inflight.pop(0))
        for d in set(devices):
            torch.cuda.synchronize(torch.device("cuda", d))
        dt = time.perf_counter() - t0
        hp = mi.host_profile(reset=True)
        profs = []
        for sh in shards:
            sh.profile(False)
            profs.append(sh.profile_collect())
        qi = (steps - 1) % len(qs)
        sid, ssc = mi.search(qs[qi].cpu().numpy(), args.k, with_minmax=False)[:2]
        same = bool(np.array_equal(last[0], sid) and np.array_equal(last[1], ssc))
        kms = [p["total_ms"] / max(p["launches"], 1) for p in profs]
        by = max(p["bytes_per_launch"] for p in profs)
        dual = bool(shards[0].get_option("pipe_dual_scan_active" if batch <= 64 else "pipe_dual_scan_wide_active"))
        step_ms = dt / steps * 1e3
        row = {"value": batch * steps / dt, "unit": "queries/s", "batch": batch, "ms_per_step": step_ms, "two_scan_streams": dual,
               "kernel_ms_per_shard" if not dual else "kernel_lifetime_ms_per_shard": kms,
               "hbm_GBps_per_device_step": by / (step_ms * 1e-3) / 1e9, "frac_step_of_8TBps_per_device": by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "host_collect_ms_per_step": t_collect / max(steps - 2, 1) * 1e3, "caller_enqueue_us_per_step": t_enq / steps * 1e6, "algorithmic_bytes_per_launch": by,
               "host_side": hp, "last_pipelined_batch_equals_synchronous_search": same}
        if not dual:
            row["frac_of_8TBps_kernel"] = by / (max(kms) * 1e-3) / 1e9 / HBM_PEAK_GBS if max(kms) else 0.0
        return row, qs, last

    head, qs, last = run(args.batch, args.steps, args.warmup, 4321)
    out = {"metric": "top-k queries/sec", "value": head["value"], "unit": "queries/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "data_note": "corpus rows drawn by " + _generator_note(),
           "config": {"workload": f"brute-force top-{args.k} over {args.rows} x {args.dim} {args.dtype} rows, batch {args.batch} (north_star target config; corpus fixed, "
                                  f"row-sharded over {n} device(s) driven from ONE process)",
                      "rows": args.rows, "dim": args.dim, "batch": args.batch, "k": args.k, "process_model": f"one process, {n} shard(s) on devices {devices} (MultiDeviceIndex)",
                      "sharding": f"rows/{n}", "shard_rows": mi.shard_rows(), "device": info["name"], "n_cu": info["n_cu"],
                      "exchange": "host-mapped: every shard's merge kernel writes its [B, k] candidates into pinned host memory, host-side final merge (no collective)"},
           "roofline": {"bound": "hbm", "achieved": head["hbm_GBps_per_device_step"] if head["two_scan_streams"] else head["algorithmic_bytes_per_launch"] / (max(head["kernel_ms_per_shard"]) * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel": "scan_kernel (fused MFMA scan + top-k), per device",
                        "achieved_is": "algorithmic bytes / step time" if head["two_scan_streams"] else "algorithmic bytes / HIP-event time of the slowest shard's scan",
                        "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"], "rows_per_gpu": max(mi.shard_rows())},
           "verified": {"last_pipelined_batch_equals_synchronous_search": head["last_pipelined_batch_equals_synchronous_search"]},
           "cpu_baseline": None, "single_process": True, "headline_detail": head}
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    ok = head["last_pipelined_batch_equals_synchronous_search"]
    if not args.no_extra and args.batch != 256 and args.dim in (768, 1024):
        c3, _, _ = run(256, max(10, args.steps // 2), 3, 8765)
        out["extra"] = {"config3_batch256": c3}
        out["verified"]["batch256_last_pipelined_batch_equals_synchronous_search"] = c3["last_pipelined_batch_equals_synchronous_search"]
        ok = ok and c3["last_pipelined_batch_equals_synchronous_search"]
    mi.close()
    if not args.no_extra:
        # the corpus encode over the same devices from the same ONE process: a replica of the layer stack per device (`embedding_devices`;
        # with --share-device: logical replicas on cuda:0), bucketing windows dealt round them, rows gathered device to device
        try:
            from tools import bench_extras as bx
            torch.cuda.empty_cache()
            e1 = bx.encode_breakdown(torch, dev0, "base", "bf16", 512, parity=False)[0]
            er = bx.encode_breakdown(torch, dev0, "base", "bf16", 512 * max(1, min(n, 4)), parity=False, devices=sorted(set(devices)), replicas=n)[0]
            keep = ("value", "chunks", "forward_only_chunks_per_s", "tokenizer_only_chunks_per_s", "encode_replicas", "gelu_path")
            out.setdefault("extra", {})["corpus_embed_bf16_replicas"] = {"one_replica": {k_: e1.get(k_) for k_ in keep}, "replicas": {k_: er.get(k_) for k_ in keep},
                                                                         "speedup": er["value"] / e1["value"]}
            out["config"]["x_corpus_embed_bf16_chunks_per_s"] = er["value"]
            out["config"]["x_corpus_embed_bf16_one_replica_chunks_per_s"] = e1["value"]
        except Exception as e:      # noqa: BLE001
            out.setdefault("extra", {})["corpus_embed_bf16_replicas"] = {"error": repr(e)[:300]}
    emit(out, side_path=None)
    if not ok:
        raise SystemExit("bench: pipelined outputs differ from the synchronous search of the same batch")


def gen_rows_dev_multi(torch, args, devices, per, blk):
    """The bench corpus block by block, every block generated ON the device whose shard takes (most of) it."""
    for b0 in range(0, args.rows, blk):
        dev = torch.device("cuda", devices[min(len(devices) - 1, b0 // per)])
        for x in gen_rows_dev(torch, b0, min(b0 + blk, args.rows), args.dim, dev, block=blk):
            yield x


def single_process_leg(args):
    """N > 1 under a launcher, rank 0, after the one-rank-per-GPU measurement: the same workload through the single-process
    index in a CHILD process (own HIP context; a hang is cut off by the timeout and reported, never inherited)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--single-process", "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--rows", str(args.rows), "--dim", str(args.dim), "--batch", str(args.batch), "--k", str(args.k), "--dtype", args.dtype,
           "--query-batches", str(args.query_batches)]
    if args.share_device:
        cmd.append("--share-device")
    if args.no_extra:
        cmd.append("--no-extra")
    if args.survey_rng:
        cmd.append("--survey-rng")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                             "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        try:
            so, se = p.communicate(timeout=args.single_process_timeout)
        except subprocess.TimeoutExpired:
            p.kill()                                   # this exact child
            p.communicate()
            return {"error": f"timeout after {args.single_process_timeout:.0f} s"}
        if p.returncode != 0:
            return {"error": f"exit code {p.returncode}", "stderr_tail": se[-600:]}
        d, side = parse_emitted(so)
        keep = {k_: d.get(k_) for k_ in ("value", "ms_per_step", "verified", "n_gpus")}
        keep["headline_detail"] = side.get("headline_detail")
        cf = side.get("config_full") or {}
        keep["process_model"] = cf.get("process_model", d["config"]["process_model"])
        keep["exchange"] = cf.get("exchange", d["config"]["exchange"])
        keep["config3_batch256"] = (side.get("extra") or {}).get("config3_batch256")
        keep["corpus_embed_bf16_replicas"] = (side.get("extra") or {}).get("corpus_embed_bf16_replicas")
        keep["wall_seconds"] = time.perf_counter() - t0
        return keep
    except Exception as e:      # the headline line must still print
        return {"error": repr(e)[:300]}


def _with_timeout(fn, seconds):
    """Run fn() on a helper thread; True if it returned within `seconds` (False: it is still stuck — e.g. in a collective whose
    peer is gone — or it raised)."""
    import threading
    box = []
    def _run():
        try:
            fn()
            box.append(True)
        except Exception:       # noqa: BLE001
            box.append(False)
    th = threading.Thread(target=_run, daemon=True)
    th.start()
    th.join(seconds)
    return bool(box and box[0])


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import torch
    have = torch.cuda.device_count()
    if args.share_device:
        if have < 1:
            raise SystemExit("bench.py --share-device needs one visible MI355X; comorag_amd has no CPU fallback")
        if args.backend != "gloo":
            raise SystemExit("bench.py --share-device: ranks sharing one GPU cannot form an RCCL group; add --backend gloo")
    elif have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s) (torch.cuda.device_count()); one rank per GPU is required "
                         f"(use --backend gloo --share-device to rehearse the multi-rank flow on one GPU)")
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    global SURVEY_RNG
    args = parse()
    SURVEY_RNG = bool(args.survey_rng)
    if args.single_process:
        return single_process_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; comorag_amd has no CPU fallback")
    if args.share_device and args.backend != "gloo":
        raise SystemExit("--share-device needs --backend gloo")
    if world > 1 and not args.share_device and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: only {torch.cuda.device_count()} GPU(s) visible; one rank per GPU is required")
    device = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(device)
    ctl = "cuda" if args.backend == "nccl" else "cpu"          # where the control-plane tensors of this backend live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from comorag_amd import _lib as L
    info = L.device_info(device.index or 0)
    # exchange bindings: batch 64 through --exchange, batch 256 through the other one, so that both get hardware time on a
    # multi-GPU node; on a gloo group only the torch binding exists (RCCL needs one device per rank)
    ex64 = args.exchange if args.backend == "nccl" else "torch"
    ex256 = ({"torch": "cabi", "cabi": "torch"}[ex64]) if args.backend == "nccl" else "torch"
    args.exchange = ex64

    qs = make_queries(torch, max(1, args.query_batches), args.batch, args.dim, device, 4321)
    qh = qs[0].cpu().numpy()

    # fp32 host copy of the corpus for the CPU legs (rank 0, N=1 only): needs rows*dim*4 bytes + slack
    host = None
    want_cpu = rank == 0 and world == 1 and not args.no_extra and not args.no_cpu_baseline and not args.only_config3
    if want_cpu:
        try:
            import psutil
            need = args.rows * args.dim * 4
            if psutil.virtual_memory().available > need + (16 << 30):
                host = np.empty((args.rows, args.dim), dtype=np.float32)
        except Exception:
            host = None
    sh = build_shard(torch, args, args.rows, rank, world, device, host=host, timing=world > 1)
    # >= 20 timed scan launches for `roofline`: every 4th launch by default, more often when steps x repeats is small
    every = max(1, min(PROFILE_EVERY, (args.steps * max(1, args.repeats)) // 20))
    sampler = ClockSampler(_pci_address(torch, device))
    dt, prof, last, qi, dts = run_steps(torch, dist, sh, qs, args.k, args.steps, args.warmup, world, device, every=every, ctl=ctl, repeats=args.repeats, sampler=sampler)
    clock = sampler.summary()
    gpu_ids, gpu_sc, same = verify_last_batch(sh, last, qs[qi].cpu().numpy(), args.k)
    if qi != 0:                 # recall is computed for batch 0 (the CPU ranking of one batch is the expensive part)
        gpu_ids = sh.search(qh, args.k)[0]
    head = summarise(args.batch, args.steps, dt, prof, len(sh), args.dim, dual=bool(sh.local.get_option("pipe_dual_scan_active")), dts=dts)
    generator = _generator_note()
    out = {
        "metric": "top-k queries/sec", "value": head["value"], "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "data_note": "corpus rows drawn by " + generator,
        # the final line keeps `config` / `roofline` / `cpu_baseline` / `verified`; the keys below the contract's travel on the EXTRA line
        "ms_per_step_all": head.get("ms_per_step_all", [head["ms_per_step"]]),
        "ms_per_step_min": head.get("ms_per_step_min", head["ms_per_step"]), "ms_per_step_max": head.get("ms_per_step_max", head["ms_per_step"]),
        "timing_note": f"the timed region (exactly {args.steps} steps between barrier + synchronize, max over ranks) ran {len(dts)} times back to back after one "
                       "warm-up; value / ms_per_step are the MEDIAN region, min / max over the regions beside them",
        "config": {"workload": f"brute-force top-{args.k} over {args.rows} x {args.dim} {args.dtype} rows, batch {args.batch}, row-sharded over {world} GPU(s) "
                               "(north_star target config)",
                   "rows": args.rows, "dim": args.dim, "batch": args.batch, "k": args.k, "query_batches_rotated": len(qs),
                   "sharding": f"rows/{world}", "device": info["name"], "n_cu": info["n_cu"],
                   "repeats": len(dts), "value_min": head.get("value_min", head["value"]), "value_max": head.get("value_max", head["value"]),
                   "generator": "numpy default_rng([1234, blk]), SURVEY 8(d) literally" if SURVEY_RNG else "torch device randn per 250K-row block, rows L2-normalised (SURVEY 8(d) distribution)",
                   "exchange": "none (1 shard)" if world == 1 else f"one packed-u64 all-gather per batch ({ex64} binding over {args.backend}"
                                                                   f"{', ranks share cuda:0, keys staged through the host' if args.share_device else ''}) + device key merge"},
        "roofline": {"bound": "hbm", "achieved": head["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac_of_8TBps"],
                     "frac_of_achievable_6290": head["hbm_GBps"] / HBM_ACHIEVABLE_GBS, "traffic": None,
                     "kernel": "scan_kernel (fused MFMA scan + top-k)", "kernel_ms": head["kernel_ms"], "two_scan_streams": head["two_scan_streams"],
                     "achieved_is": "algorithmic bytes / step time (launches overlap: no per-launch duration)" if head["two_scan_streams"] else "algorithmic bytes / HIP-event time of the scan on its stream",
                     "algorithmic_bytes_per_launch": prof["bytes_per_launch"], "launches_timed": prof["launches"], "timed_every": every,
                     "rows_per_gpu": len(sh), "clock": clock},
        "verified": {"last_pipelined_batch_equals_synchronous_search": same},
    }
    if not same:
        out["verified"]["error"] = "pipelined outputs differ from the synchronous search of the same batch"

    # BASELINE config 3 as written: the same (sharded) corpus, batch 256 — one corpus pass of the wide kernel per batch
    # (multi-rank: this section holds the FIRST executions of the library's own RCCL communicator on real hardware — it runs under a
    #  watchdog, so that a collective that never returns costs the extras, not the headline line above)
    c3 = None
    same_w = True
    q256 = None
    shw = sh
    section_error = []

    def _after_headline():
        nonlocal c3, same_w, q256, shw
        torch.cuda.set_device(device)
        if not args.no_extra and args.batch != 256 and args.dim in (768, 1024):
            q256 = make_queries(torch, max(1, min(args.query_batches, 3)), 256, args.dim, device, 8765)
            steps_w = max(10, args.steps // 2)
            shw = sh.view(ex256, timing=world > 1) if world > 1 else sh
            samp_w = ClockSampler(_pci_address(torch, device))
            dtw, profw, lastw, qiw, dtsw = run_steps(torch, dist, shw, q256, args.k, steps_w, 3, world, device, ctl=ctl, repeats=min(3, max(1, args.repeats)), sampler=samp_w)
            _, _, same_w = verify_last_batch(shw, lastw, q256[qiw].cpu().numpy(), args.k)
            c3 = summarise(256, steps_w, dtw, profw, len(sh), args.dim, dual=bool(sh.local.get_option("pipe_dual_scan_wide_active")), dts=dtsw)
            c3["kernel"] = "scan_wide_kernel (256 queries resident in registers, LDS-DMA corpus ring)"
            c3["clock"] = samp_w.summary()
            c3["frac_of_2500TF_bf16"] = c3["mfma_TFLOPs"] / MFMA_BF16_PEAK_TFLOPS
            c3["last_pipelined_batch_equals_synchronous_search"] = same_w
            c3["exchange_binding"] = ex256 if world > 1 else None
            if world > 1 and args.backend == "nccl":
                try:
                    c3["rccl_ranks_seen"] = (shw if ex256 == "cabi" else sh).comm_info()["rccl_ranks_seen"]
                except Exception as e:
                    c3["rccl_ranks_seen"] = repr(e)[:200]
        if world > 1:
            out["verified"]["batch256_last_pipelined_batch_equals_synchronous_search"] = same_w
            out["exchange_bindings"] = {"batch64": ex64, "batch256": ex256 if c3 else None, "backend": args.backend, "share_device": bool(args.share_device),
                                        "torch_world_size": dist.get_world_size()}
            if args.backend == "nccl":
                try:        # ranks RCCL itself reports for the library's own communicator (ncclCommCount)
                    cv = sh if ex64 == "cabi" else (shw if c3 else sh.view("cabi"))
                    out["exchange_bindings"]["rccl_ranks_seen"] = cv.comm_info()["rccl_ranks_seen"]
                except Exception as e:
                    out["exchange_bindings"]["rccl_ranks_seen"] = repr(e)[:200]
            mine = {"rank": rank, "rows": len(sh), "device": str(device), "batch64": {k_: head.get(k_) for k_ in ("kernel_ms", "kernel_lifetime_ms", "exchange_ms", "merge_ms", "ms_per_step")},
                    "batch256": {k_: c3.get(k_) for k_ in ("kernel_ms", "kernel_lifetime_ms", "exchange_ms", "merge_ms", "ms_per_step")} if c3 else None}
            if not args.no_extra:
                # the encode half of the metric shards as the rows do (chunks are independent; the reference: device_map="auto",
                # BGEEmbedding.py:77): every rank encodes its own chunks on its own GPU at the same time — per rank and summed
                try:
                    from tools import bench_extras as bx
                    dist.barrier()
                    e_ = bx.encode_breakdown(torch, device, "base", "bf16", 256, parity=False)[0]
                    mine["corpus_embed_bf16"] = {k_: e_.get(k_) for k_ in ("value", "forward_only_chunks_per_s", "tokenizer_only_chunks_per_s", "chunks", "gelu_path", "encoder_path")}
                except Exception as e:      # noqa: BLE001
                    mine["corpus_embed_bf16"] = {"error": repr(e)[:200]}
            box = [None] * world
            dist.all_gather_object(box, mine)
            out["per_rank"] = box
            rates = [(p_.get("corpus_embed_bf16") or {}).get("value") for p_ in box]
            if all(isinstance(v_, float) for v_ in rates):
                out["config"]["x_corpus_embed_bf16_chunks_per_s_sum_over_ranks"] = float(sum(rates))
                out["config"]["x_corpus_embed_bf16_chunks_per_s_min_rank"] = float(min(rates))

    if world > 1:
        import threading
        def _guarded():
            try:
                _after_headline()
            except Exception as e:      # noqa: BLE001
                section_error.append(repr(e)[:400])
        th = threading.Thread(target=_guarded, daemon=True)
        th.start()
        th.join(args.extras_timeout)
        if th.is_alive() or section_error:
            why = section_error[0] if section_error else f"no return after {args.extras_timeout:.0f} s (a collective of the batch-256 / per-rank section never completed)"
            out["extra"] = {"config3_batch256": {"error": why}}
            out["multi_rank_extras"] = "abandoned: " + why
            if rank == 0:
                emit(out, side_path=SIDE_FILE)
            os._exit(0)                 # ranks may sit in a dead collective: no further collectives, no clean teardown
    else:
        _after_headline()
    rows_here = len(sh)
    if world > 1 and c3 and shw is not sh:
        shw.close()
    sh.close()
    del sh
    # roofline.traffic: HBM bytes per launch of the scan kernel from the PMC counters, measured BY THIS RUN — the shard is freed,
    # bench.py re-runs itself for a few steps under rocprofv3 (FETCH_SIZE and WRITE_SIZE passes, tools/pmc_traffic.py); the
    # committed measurement of an earlier round is only the fallback when rocprofv3 is missing or a pass fails
    if rank == 0 and world == 1 and not args.no_pmc and not args.share_device:
        try:
            from tools import pmc_traffic
            torch.cuda.empty_cache()
            argv = ["--rows", str(args.rows), "--dim", str(args.dim), "--batch", str(args.batch), "--k", str(args.k), "--dtype", args.dtype,
                    "--query-batches", str(args.query_batches)] + [x for o in args.index_option for x in ("--index-option", o)] + (["--survey-rng"] if args.survey_rng else [])
            tr = pmc_traffic.measure(argv)
        except Exception as e:      # noqa: BLE001
            tr = {"error": repr(e)[:300]}
        if "error" not in tr:
            out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
            out["roofline"]["traffic_over_algorithmic"] = tr["traffic_bytes_per_launch"] / prof["bytes_per_launch"]
            out["roofline"]["traffic_source"] = "pmc passes of this run (rocprofv3), FETCH x 2 + WRITE"
            out["roofline"]["traffic_launches"] = tr["raw"]["FETCH_SIZE"]["launches"]
            out["roofline"]["traffic_kernel_us_under_pmc"] = tr["raw"]["FETCH_SIZE"]["avg_kernel_us"]
            out["pmc_this_run"] = tr
        else:
            out["roofline"]["traffic_error"] = tr["error"][:110]
    if out["roofline"]["traffic"] is None:
        for prof_file in ("r5_pmc_hbm_traffic.json", "r4_pmc_hbm_traffic.json", "r3_pmc_hbm_traffic.json"):
            prof_file = os.path.join(ROOT, "profiles", prof_file)
            if os.path.exists(prof_file):
                pj = json.load(open(prof_file))
                w = pj.get("workload", {})
                if (w.get("rows"), w.get("dim"), w.get("dtype"), w.get("batch"), w.get("k")) == (rows_here, args.dim, args.dtype, args.batch, args.k):
                    out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = f"FALLBACK profiles/{os.path.basename(prof_file)} (earlier round, NOT this run)"
                    break
    out["cpu_baseline"] = None
    extra = {}
    if c3 is not None:
        extra["config3_batch256"] = c3
    if rank == 0 and world == 1 and not args.no_extra and not args.only_config3:
        q = qs
        from comorag_amd.sharded import ShardedIndex
        # BASELINE config 2 (1 M rows) and one 8-GPU shard of config 3 (1.25 M rows) on this GPU
        for name, rows2, batches in (("config2", min(args.rows, 1_000_000), (args.batch,)), ("config3_one_of_8_shards", min(args.rows, 1_250_000), (args.batch, 256))):
            sh2 = build_shard(torch, args, rows2, 0, 1, device)
            for b2 in batches:
                if b2 != args.batch and q256 is None:
                    continue
                qq = q if b2 == args.batch else q256
                steps2 = max(args.steps, 100)
                dt2, prof2, _, _, dts2 = run_steps(torch, dist, sh2, qq, args.k, steps2, args.warmup, 1, device, repeats=min(3, max(1, args.repeats)))
                extra[f"{name}_{rows2}_rows_batch{b2}"] = summarise(b2, steps2, dt2, prof2, rows2, args.dim, dts=dts2,
                                                                     dual=bool(sh2.local.get_option("pipe_dual_scan_active" if b2 <= 64 else "pipe_dual_scan_wide_active")))
            if name == "config2":
                try:
                    # (every size on an index of its own, as tools/latency.py measures it: the pipelined shard above still has its scans' event timing
                    # switched on, and two event records around every n-th scan are part of such a call)
                    lat = single_query_latency(torch, args, device, sizes=(6, 1000, 10_000, 100_000, rows2))
                    extra["single_query_latency"] = lat
                except Exception as e:
                    extra["single_query_latency"] = {"error": repr(e)[:300]}
                extra["host_buffer_api"] = host_api_rate(torch, sh2, qh, args.k)
                extra["host_buffer_api"]["note"] = f"PCIe-inclusive: {rows2} rows, H2D queries + D2H results + sync per call"
                try:        # complete ranking of one query (dense_passage_retrieval's all-N return): scan + device radix sort + D2H
                    sh2.local.sorted_scores(qh[:1])
                    t0 = time.perf_counter()
                    for i in range(10):
                        sh2.local.sorted_scores(qh[i % len(qh):i % len(qh) + 1])
                    dtr = (time.perf_counter() - t0) / 10
                    x1 = sh2.local.scores(qh[:1])[0]
                    t0 = time.perf_counter()
                    np.argsort(x1)[::-1]
                    extra["full_ranking_one_query"] = {"rows": rows2, "ms_per_query": dtr * 1e3, "host_argsort_ms": (time.perf_counter() - t0) * 1e3,
                                                       "note": "PCIe-inclusive: N int64 ids + N fp32 scores copied back"}
                except Exception as e:
                    extra["full_ranking_one_query"] = {"error": repr(e)[:300]}
            sh2.close()
        s64 = extra.get(f"config3_one_of_8_shards_{min(args.rows, 1_250_000)}_rows_batch{args.batch}")
        s256 = extra.get(f"config3_one_of_8_shards_{min(args.rows, 1_250_000)}_rows_batch256")
        if s64 and s256 and c3:
            extra["projected_8gpu"] = {"projected": True,
                                       "basis": "step time of ONE 1.25 M-row shard measured on this GPU vs the 10 M-row step above; the exchange (10-40 KiB per rank) "
                                                "runs on the post stream under the next scan and is not included",
                                       "batch64_speedup": head["ms_per_step"] / s64["ms_per_step"], "batch256_speedup": c3["ms_per_step"] / s256["ms_per_step"]}
        # BASELINE configs 4 / 5, SURVEY 8 f1 / f4, and the encode leg of the metric with its breakdown (tools/bench_extras.py)
        from tools import bench_extras as bx
        rows = (("config4_probe_loop", lambda: bx.config4_probe_loop(torch, device, dim=args.dim, dtype=args.dtype, rows0=min(2_000_000, max(args.rows, 200_000)), k=args.k)),
                ("corpus_embed", lambda: bx.encode_breakdown(torch, device, "base", "auto", 128)[0]),
                ("corpus_embed_bf16", lambda: bx.encode_breakdown(torch, device, "base", "bf16", 1024)[0]),
                ("corpus_embed_bf16_tokenizer_processes", lambda: bx.encode_breakdown(torch, device, "base", "bf16", 1024, tok_processes=-1)[0]),
                ("config5_bge_large_fp16_encode_search_rescore", lambda: bx.config5_encode_search_rescore(torch, device)),
                ("f1_synonymy_selfjoin", lambda: bx.f1_selfjoin(torch, device, dim=args.dim)),
                ("f4_dpr_seeded_ppr", lambda: bx.f4_ppr(torch, device)))
        for name, fn in rows:
            try:
                t0 = time.perf_counter()
                extra[name] = fn()
                extra[name]["bench_seconds"] = time.perf_counter() - t0
            except Exception as e:  # the headline line must still print
                extra[name] = {"error": repr(e)[:400]}
            torch.cuda.empty_cache()
        # the CPU legs come LAST: their 64-128 BLAS / OpenMP threads leave the host busy for a while, and the encode rows above are
        # host-sensitive (tokenizer threads feeding the forward: end to end 0.90-0.91 of forward-only behind the CPU legs, 0.93-0.97 in front)
        if not args.no_cpu_baseline:
            if host is None:                    # not enough host RAM for the full fp32 copy: first 1 M rows, scaled
                n1 = min(args.rows, 1_000_000)
                host = np.empty((n1, args.dim), dtype=np.float32)
                at = 0
                for blk in gen_rows_dev(torch, 0, n1, args.dim, device):
                    host[at:at + len(blk)] = blk.cpu().numpy(); at += len(blk)
                full = n1 == args.rows
                sh3 = build_shard(torch, args, n1, 0, 1, device)
                ids_for_recall = sh3.local.search(qh, args.k)[0]
                sh3.close()
            else:
                full, ids_for_recall = True, gpu_ids
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds, host, qh, ids_for_recall, full)
        del host
    if extra:
        out["extra"] = extra
        # the secondary numbers that matter, flat, where the driver's parser keeps them (it drops `extra`)
        def _get(d_, *path):
            for p_ in path:
                d_ = d_.get(p_) if isinstance(d_, dict) else None
            return d_
        flat = {"config3_batch256_kernel_ms": _get(extra, "config3_batch256", "kernel_ms"), "config3_batch256_qps": _get(extra, "config3_batch256", "value"),
                "config3_sclk_mhz": _get(extra, "config3_batch256", "clock", "sclk_mhz_median"), "config3_power_w": _get(extra, "config3_batch256", "clock", "power_w_mean"),
                "config3_frac_of_2500TF": _get(extra, "config3_batch256", "frac_of_2500TF_bf16"),
                "config2_1M_rows_ms_per_step": _get(extra, f"config2_{min(args.rows, 1_000_000)}_rows_batch{args.batch}", "ms_per_step"),
                "config2_1M_rows_qps": _get(extra, f"config2_{min(args.rows, 1_000_000)}_rows_batch{args.batch}", "value"),
                "shard_1p25M_rows_batch256_ms_per_step": _get(extra, f"config3_one_of_8_shards_{min(args.rows, 1_250_000)}_rows_batch256", "ms_per_step"),
                "corpus_embed_bf16_chunks_per_s": _get(extra, "corpus_embed_bf16", "value"),
                "corpus_embed_bf16_forward_only_chunks_per_s": _get(extra, "corpus_embed_bf16", "forward_only_chunks_per_s"),
                "corpus_embed_bf16_end_to_end_over_forward_only": _get(extra, "corpus_embed_bf16", "end_to_end_over_forward_only"),
                "config5_bge_large_fp16_chunks_per_s": _get(extra, "config5_bge_large_fp16_encode_search_rescore", "encode", "value"),
                "config5_end_to_end_over_forward_only": _get(extra, "config5_bge_large_fp16_encode_search_rescore", "encode", "end_to_end_over_forward_only"),
                "single_query_latency_1M_rows_us": _get(extra, "single_query_latency", "rows", str(min(args.rows, 1_000_000))),
                "config4_search_us_per_call": _get(extra, "config4_probe_loop", "search_us_per_call"),
                "config4_call_frac_of_hbm": _get(extra, "config4_probe_loop", "frac"),
                "f1_selfjoin_bf16_s": _get(extra, "f1_synonymy_selfjoin", "bf16", "threshold_search_whole_join_s"),
                "f1_selfjoin_bf16_frac_of_2500TF": _get(extra, "f1_synonymy_selfjoin", "bf16", "frac"),
                "f1_selfjoin_bf16_ids_download_s": _get(extra, "f1_synonymy_selfjoin", "bf16", "ids_download_s"),
                "f4_ppr_comorag_scale_us": _get(extra, "f4_dpr_seeded_ppr", "comorag_scale", "fused_us_per_query"),
                "f4_ppr_1M_passages_us": _get(extra, "f4_dpr_seeded_ppr", "at_1M_passages", "fused_us_per_query"),
                "attention_us_per_layer_bf16": _get(extra, "corpus_embed_bf16", "attention_us_per_layer"),
                "encoder_gelu_path": _get(extra, "corpus_embed_bf16", "gelu_path"),
                "encoder_parity_min_row_cosine": _get(extra, "corpus_embed_bf16", "parity_vs_fp32_oracle", "min_row_cosine_vs_fp32_oracle")}
        out["config"].update({f"x_{k_}": v_ for k_, v_ in flat.items() if v_ is not None})
    left_cleanly = True
    if world > 1:
        # every rank has freed its shard; the ranks leave the group BEFORE rank 0 starts the single-process leg, so that no
        # collective is pending while ONE child process takes all the GPUs (under a watchdog: a peer that died in the section
        # above must not keep the line from being printed)
        def _leave():
            torch.cuda.set_device(device)
            torch.cuda.synchronize(device)
            dist.barrier()
            dist.destroy_process_group()
        left_cleanly = _with_timeout(_leave, 120.0)
        if rank == 0 and not args.no_single_process_leg and left_cleanly:
            torch.cuda.empty_cache()
            out["single_process"] = single_process_leg(args)
            if isinstance(out["single_process"].get("value"), float):
                out["config"]["x_single_process_qps"] = out["single_process"]["value"]
            er_ = ((out["single_process"].get("corpus_embed_bf16_replicas") or {}).get("replicas") or {}).get("value")
            if isinstance(er_, float):
                out["config"]["x_single_process_corpus_embed_bf16_chunks_per_s"] = er_
        elif rank == 0 and not left_cleanly:
            out["single_process"] = {"error": "skipped: the ranks did not leave the process group within 120 s"}
    if rank == 0:
        emit(out, side_path=SIDE_FILE)
    if not left_cleanly:
        os._exit(0 if (same and same_w) else 1)
    if not (same and same_w):
        raise SystemExit("bench: pipelined outputs differ from the synchronous search of the same batch")


if __name__ == "__main__":
    main()
