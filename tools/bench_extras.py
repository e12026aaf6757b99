"""The rows of bench.py's `extra` that time BASELINE configs 4 and 5 and SURVEY §8 f1 / f4 (rank 0, N = 1 only).

Every function returns one JSON-able dict with the measured rates, a `frac` of the roofline that bounds the dominant
kernel of that row, a parity bit checked inside the run, and a `cpu` twin: the oracle's restatement of the reference code
(or the third-party routine the reference calls) timed on this box's host cores over a bounded sample.  The oracle is used
here as bench.py uses it: as the checker and as the `cpu_baseline` leg, never as the thing measured.
Reference sites: utils/memory_utils.py:188-235,294-300 (config 4), embedding_model/BGEEmbedding.py:131-185 + rerank.py
(config 5), ComoRAG.py:670-712 + utils/embed_utils.py:8-97 (f1), ComoRAG.py:1034-1105 (f4).
"""
from __future__ import annotations

import time

import numpy as np

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 / fp16 peak
F32_PEAK_TFLOPS = 157.0            # same guide: fp32 matrix rate


def _unit_rows_dev(torch, n, dim, device, seed, block=250_000):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for b in range(0, n, block):
        x = torch.randn((min(block, n - b), dim), generator=g, device=device, dtype=torch.float32)
        yield (x / x.norm(dim=1, keepdim=True)).contiguous()


def _median_us(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t) * 1e6)


# ------------------------------------------------------------------------------------------------ config 4
def config4_probe_loop(torch, device, dim=768, dtype="bf16", rows0=2_000_000, k=20, cycles=5, probes=8, cpu_nodes=20_000):
    """BASELINE config 4: 5 reasoning cycles x (ONE B = 8 search over the 2 M-chunk memory pool, k = 20, then an append of
    25 rows [3 nodes x 8 probes + 1 fusion]); afterwards one 65 536-row burst that forces a capacity doubling.  The index is
    created with room for the cycles' small appends only, so the growth path (hipMemcpyAsync into a 2x allocation) is what
    the burst times.  Every appended row is searched for right after its append: it must come back first with its
    append-order id."""
    from comorag_amd.index import DenseIndex
    idx = DenseIndex(dim, dtype, device=device.index or 0, capacity_hint=rows0 + 4096)     # room for the cycles' 25-row appends, not for the burst
    for blk in _unit_rows_dev(torch, rows0, dim, device, 4001):
        idx.append_dev(blk)
    torch.cuda.synchronize(device)
    rng = np.random.default_rng(4002)
    def unit(m):
        x = rng.standard_normal((m, dim)).astype(np.float32)
        return x / np.linalg.norm(x, axis=1, keepdims=True)
    q = unit(probes)
    for _ in range(3):
        idx.search(q, k)
    t_search, t_append, found = [], [], True
    for c in range(cycles):
        t0 = time.perf_counter(); idx.search(q, k); t_search.append(time.perf_counter() - t0)
        new = unit(25)
        n_before = len(idx)
        t0 = time.perf_counter(); idx.append(new); t_append.append(time.perf_counter() - t0)
        ids, sc = idx.search(new[:probes], 1)[:2]
        found &= ids[:, 0].tolist() == list(range(n_before, n_before + probes))
        q = unit(probes)
    # the scan's own time by HIP events, on calls of their own: two event records around the scan are ~10 us of a timed call
    idx.profile(True)
    for _ in range(10):
        idx.search(q, k)
    prof = idx.profile_collect()
    idx.profile(False)
    burst = unit(65_536)
    cap_before = idx.device_bytes
    n_before = len(idx)
    t0 = time.perf_counter(); idx.append(burst); t_burst = time.perf_counter() - t0
    grew = idx.device_bytes > cap_before
    ids = idx.search(burst[[0, 65_535]], 1)[0]
    found &= ids[:, 0].tolist() == [n_before, n_before + 65_535]
    scan_bytes = rows0 * dim * 2 + probes * dim * 4 + probes * k * 12
    kernel_ms = prof["total_ms"] / max(prof["launches"], 1)
    us = float(np.median(t_search) * 1e6)
    out = {"rows": rows0, "dim": dim, "dtype": dtype, "probes_per_search": probes, "k": k, "cycles": cycles,
           "search_us_per_call": us, "append_25_rows_us": float(np.median(t_append) * 1e6),
           "append_65536_rows_ms": t_burst * 1e3, "burst_grew_capacity": bool(grew), "burst_GBps_host_to_index": 65_536 * dim * 4 / t_burst / 1e9,
           "appended_rows_found_first_with_dense_ids": bool(found),
           "scan_kernel_ms": kernel_ms, "scan_kernel_launches_timed": prof["launches"],
           "frac": scan_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_kernel": (prof["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms else None,
           "frac_of": "8 TB/s HBM; `frac` = algorithmic bytes of one B = 8 search / the whole synchronous call (PCIe queries in, results out, Python wrapper), `frac_kernel` = the same bytes / the main scan's HIP-event time",
           "note": "synchronous host-buffer API, as MemoryPool.retrieve_similar_nodes / add_node issue it (utils/memory_utils.py:188-235,294-300)"}
    idx.close()
    # CPU twin: the reference's python-loop cosine over the pool (oracle restatement), bounded sample, linear in the pool size
    try:
        from oracle import retrieval_np as orc
        nodes = list(unit(cpu_nodes))
        t0 = time.perf_counter(); orc.retrieve_similar_nodes(nodes, q[0], 0.5); dt = time.perf_counter() - t0
        out["cpu"] = {"what": "oracle.retrieve_similar_nodes = utils/memory_utils.py:213-235 (python-loop cosine + stable sort), ONE probe",
                      "sample_nodes": cpu_nodes, "ms_per_probe_sample": dt * 1e3, "ms_per_probe_scaled_to_pool": dt * 1e3 * rows0 / cpu_nodes,
                      "scaled": f"x{rows0 / cpu_nodes:g} (the loop is linear in the pool size)", "cores": 1}
    except Exception as e:
        out["cpu"] = {"error": repr(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------------ config 5
def bert_flops_per_chunk(hidden, layers, tokens, inter_mult=4):
    # per token and layer: QKV + out projections 4 h^2 MACs, FFN 2 * inter_mult h^2 MACs, attention scores + weighted sum
    # 2 T h MACs; x 2 flops per MAC
    return 2.0 * tokens * layers * ((4 + 2 * inter_mult) * hidden * hidden + 2.0 * tokens * hidden)


def encoder_parity(torch, em, chunks, n=32, peak=None):
    """The product encode path (`em.batch_encode`: tokenise -> 16-bit layer stack -> HIP pool + L2-norm) against the oracle's fp32
    restatement of the reference's batch_encode (oracle/encode_torch.py, CPU) on the SAME 16-bit-representable weights, for n
    RAGGED chunks (17 ... ~512 word pieces).  north_star's bar for 16-bit arithmetic: cosine scores within 1e-3 —
    reported: the worst row cosine between the two embeddings and the worst difference of any pairwise score."""
    import copy
    from oracle import encode_torch as enc_o
    ragged = [" ".join(c.split()[:17 + (i * 16) % 500]) for i, c in enumerate(chunks[:n])]
    model32 = copy.deepcopy(em.embedding_model).float().cpu().eval()
    maxlen = int(getattr(em.embedding_model.config, "max_position_embeddings", 512))
    want = enc_o.batch_encode(model32, em.tokenizer, ragged, batch_size=8, max_length=maxlen)
    got = em.batch_encode(ragged)
    cos = (got.astype(np.float64) * want.astype(np.float64)).sum(1)
    ds = np.abs(got.astype(np.float64) @ got.astype(np.float64).T - want.astype(np.float64) @ want.astype(np.float64).T)
    out = {"rows": len(ragged), "min_row_cosine_vs_fp32_oracle": float(cos.min()), "max_abs_pairwise_score_diff": float(ds.max()),
           "max_abs_component_diff": float(np.abs(got - want).max()), "bar": "|cos - 1| <= 1e-3 and |score diff| <= 1e-3 (north_star, 16-bit)",
           "parity": bool(cos.min() >= 1.0 - 1e-3 and ds.max() <= 1e-3),
           "oracle": "oracle/encode_torch.batch_encode (BGEEmbedding.py:92-185 restated), fp32 on the host, same weights"}
    # the reference's own code in the SAME 16-bit dtype (transformers forward on the GPU + torch pooling): what 16-bit arithmetic
    # costs without any of this repo's kernels
    dev = next(em.embedding_model.parameters()).device
    plain = np.concatenate([enc_o.encode(em.embedding_model, em.tokenizer, ragged[i:i + 8], instruction=enc_o.BGE_PREFIX, max_length=maxlen).float().cpu().numpy()
                            for i in range(0, len(ragged), 8)]).astype(np.float64)
    out["transformers_same_dtype_min_row_cosine"] = float((plain * want.astype(np.float64)).sum(1).min())
    out["transformers_same_dtype_max_abs_pairwise_score_diff"] = float(np.abs(plain @ plain.T - want.astype(np.float64) @ want.astype(np.float64).T).max())
    return out


def encode_breakdown(torch, device, kind="base", dtype="bf16", n_chunks=256, batch=32, tok_processes=0, parity=True, devices=None, replicas=0):
    """Corpus-embed chunks/s end to end and where the time goes: tokenizer alone (host), forward + pool alone (device
    inputs ready), pool kernel alone; MFMA fraction of the forward from the model's matmul flops.
    devices / replicas: `embedding_devices` / `embedding_encode_replicas` — the corpus encode dealt over that many copies of the layer
    stack (one per GPU of the node; logical replicas on a one-GPU box)."""
    from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel, pool_l2norm
    from comorag_amd.utils.config_utils import BaseConfig
    from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_batch_size=batch, embedding_model_dtype=dtype, device=device.index or 0)
    cfg.embedding_tokenizer_processes = tok_processes
    if devices is not None or replicas:
        cfg.embedding_devices, cfg.embedding_encode_replicas = (list(devices) if devices is not None else None), int(replicas)
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=random_bert(kind, vocab_size=len(tok)), tokenizer=tok)
    chunks = synthetic_chunks(words, n_chunks, tokens_per_chunk=560)          # > 512 word pieces: every chunk is truncated to 512 positions
    em.batch_encode(chunks[:2 * batch])
    torch.cuda.synchronize(device)
    if tok_processes < 0 and em._tok_procs_auto:
        # -1 (opt-in; BaseConfig's default is 0 = threads) starts the tokenizer worker processes at the first corpus-sized call, in the background; the timed call below
        # is the steady state of a corpus encode, so the start-up (about a second, once per model) happens here, untimed
        em._maybe_start_tok_procs(1 << 30, 1)
        if em._tok_procs_starting is not None:
            em._tok_procs_starting.join(60.0)
    em._trace = []
    t0 = time.perf_counter(); out = em.batch_encode(chunks); torch.cuda.synchronize(device); dt_e2e = time.perf_counter() - t0
    trace, em._trace = em._trace, None
    # tokenizer alone
    t0 = time.perf_counter()
    for i in range(0, n_chunks, batch):
        em._tokenize(chunks[i:i + batch], 512)
    dt_tok = time.perf_counter() - t0
    # forward + pool alone: the path batch_encode runs (fused_bert.FusedBertLayers for 16-bit BERT encoders), and the transformers
    # forward of the same model beside it
    host_inp = em._tokenize(chunks[:batch], 512)
    inp = {k: v.to(device) for k, v in host_inp.items()}
    tokens = int(inp["input_ids"].shape[1])
    lens = host_inp["attention_mask"].numpy().sum(1).astype(np.int32)
    product = (lambda: em._fused(inp["input_ids"], lens, token_type_ids=inp.get("token_type_ids"))) if em._fused is not None else None
    plain = lambda: em.embedding_model(**inp).last_hidden_state

    def timed(fn, reps):
        for _ in range(2):
            out_ = fn()
        torch.cuda.synchronize(device)
        t0_ = time.perf_counter()
        for _ in range(reps):
            out_ = fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0_) / reps, out_

    stages = {}
    with torch.no_grad():
        dt_plain, hidden = timed(plain, 6)
        dt_fwd = dt_plain
        if product is not None:
            dt_fwd, hidden = timed(product, 6)
            fz = em._fused
            T = batch * tokens
            qkv = torch.randn((T, 3 * fz.hidden), device=device).to(fz.dtype)
            lens_dev = torch.from_numpy(lens).to(device)
            dt_att, ctx = timed(lambda: fz.attention(qkv, lens_dev, batch, tokens), 20)
            lyr = fz.layers[0]
            dt_ln, _ = timed(lambda: fz.add_layernorm(ctx, lyr[3], ctx, lyr[4], lyr[5]), 20)
            att_flops = 4.0 * float((lens.astype(np.float64) ** 2).sum()) * fz.hidden
            stages = {"attention_us_per_layer": dt_att * 1e6, "attention_TFLOPs": att_flops / dt_att / 1e12,
                      "attention_frac_of_2500TF": att_flops / dt_att / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                      "add_layernorm_us": dt_ln * 1e6, "add_layernorm_GBps": 3.0 * T * fz.hidden * 2 / dt_ln / 1e9,
                      "add_layernorm_frac_of_8TBps": 3.0 * T * fz.hidden * 2 / dt_ln / 1e9 / HBM_PEAK_GBS,
                      "transformers_forward_only_chunks_per_s": batch / dt_plain}
        dt_pool, _ = timed(lambda: pool_l2norm(hidden, inp["attention_mask"]), 20)
    h, L = em.embedding_model.config.hidden_size, em.embedding_model.config.num_hidden_layers
    flops = bert_flops_per_chunk(h, L, tokens)
    fwd_rate = batch / dt_fwd
    pool_bytes = batch * tokens * h * hidden.element_size() + batch * tokens * 8 + batch * h * 4
    res = {"model": f"BERT-{kind} shape ({L} layers, hidden {h}), random init, {dtype}", "batch": batch, "chunks": n_chunks, "tokens_per_chunk": tokens,
           "value": n_chunks / dt_e2e, "unit": "chunks/s", "embedding_dim": int(out.shape[1]),
           "tokenizer_only_chunks_per_s": n_chunks / dt_tok, "forward_only_chunks_per_s": fwd_rate,
           "pool_l2norm_us_per_batch": dt_pool * 1e6, "pool_GBps": pool_bytes / dt_pool / 1e9, "pool_frac_of_8TBps": pool_bytes / dt_pool / 1e9 / HBM_PEAK_GBS,
           "gflop_per_chunk": flops / 1e9, "forward_TFLOPs": fwd_rate * flops / 1e12, "frac": fwd_rate * flops / 1e12 / MFMA_BF16_PEAK_TFLOPS if dtype != "auto" else fwd_rate * flops / 1e12 / F32_PEAK_TFLOPS,
           "frac_of": ("2.5 PFLOP/s dense bf16/fp16 MFMA" if dtype != "auto" else "157 TFLOP/s fp32") + " for the forward alone (GEMMs: PyTorch-ROCm / hipBLASLt, by north_star's design; attention and bias + residual + LayerNorm: HIP for 16-bit BERT encoders); end-to-end = tokenizer overlapped with forward + HIP pool",
           "end_to_end_over_forward_only": (n_chunks / dt_e2e) / fwd_rate,
           "tokenizer_processes": (f"auto: {em._tok_procs_auto} worker processes" if (tok_processes < 0 and em._tok_procs is not None) else tok_processes),
           "gelu_path": getattr(em._fused, "gelu_path", None) if em._fused is not None else None,
           "encoder_path": em.encoder_path, "encode_replicas": [str(r.device) for r in em._replicas] or [str(em.device)], **stages,
           **({"parity_vs_fp32_oracle": encoder_parity(torch, em, chunks)} if parity and dtype != "auto" else {}),
           "host_ms": {"end_to_end": dt_e2e * 1e3, "waiting_for_token_ids": sum(t[0] for t in trace) * 1e3,
                       "padding_and_launching": sum(t[1] for t in trace) * 1e3, "windows": len(trace)}}
    return res, em


def config5_encode_search_rescore(torch, device, n_chunks=391, cpu_chunks=2):
    """BASELINE config 5: a 200 K-token narrative corpus = 391 chunks x 512 tokens through a BGE-large-shaped encoder in
    fp16 (batch 32), the 1024-d fp16 index with its fp32 shadow, then B = 8 searches at k = 100 and the exact fp32 re-score
    of those 100 candidates to the top 20 (the numeric stage behind rerank.py's call shape).  The corpus-scale search is a
    single-launch call (391 rows); the same search + re-score over 1 M synthetic 1024-d fp16 rows gives the scan's roofline
    fraction."""
    from comorag_amd.index import DenseIndex
    from comorag_amd.rerank import ExactRescorer  # noqa: F401  (the call shape lives there; the numeric call is index.rescore)
    enc, em = encode_breakdown(torch, device, "large", "fp16", n_chunks=n_chunks, batch=32)
    from tools.synthetic import synthetic_chunks, synthetic_wordpiece_tokenizer
    _, words = synthetic_wordpiece_tokenizer()
    chunks = synthetic_chunks(words, n_chunks, tokens_per_chunk=560)
    X = em.batch_encode(chunks)
    Q = em.batch_encode(chunks[:8])                       # queries = the first eight chunks: each must find itself first
    idx = DenseIndex(1024, "f16", device=device.index or 0, capacity_hint=n_chunks, keep_f32=True)
    idx.append(X)
    ids, sc = idx.search(Q, 100)[:2]
    rid, rsc = idx.rescore(Q, ids, 20)
    # parity inside the run: the re-scored top 20 of every query equal the fp64 ranking of ITS 100 candidates (tie-aware:
    # random-init encoders give near-identical chunk vectors), every query finds its own chunk first, and the fp16
    # candidate set holds the fp64 top 20 of the whole corpus (reported as a recall)
    from oracle import retrieval_np as orc
    exact = (Q.astype(np.float64) @ X.astype(np.float64).T)
    ok = rid[:, 0].tolist() == list(range(8))
    recall = 0.0
    for i in range(8):
        cand = ids[i][ids[i] >= 0]
        order = cand[np.lexsort((cand, -exact[i][cand]))][:20]
        try:
            orc.assert_topk_equivalent(rid[i], order, exact[i], 4e-6)
        except AssertionError:
            ok = False
        recall += len(set(np.argsort(-exact[i], kind="stable")[:20].tolist()) & set(cand.tolist())) / 20 / 8
    us_search = _median_us(lambda: idx.search(Q, 100), 30)
    us_rescore = _median_us(lambda: idx.rescore(Q, ids, 20), 30)
    idx.close()
    out = {"encode": enc, "corpus_chunks": n_chunks, "search_B8_k100_us": us_search, "rescore_100_to_20_us": us_rescore,
           "rescored_top20_equal_fp64_ranking_of_candidates_and_self_first": bool(ok), "fp64_top20_inside_fp16_top100": recall}
    # the same two calls at a size where the scan is the cost
    big = 1_000_000
    bidx = DenseIndex(1024, "f16", device=device.index or 0, capacity_hint=big, keep_f32=True)
    for blk in _unit_rows_dev(torch, big, 1024, device, 5001):
        bidx.append_dev(blk)
    torch.cuda.synchronize(device)
    rng = np.random.default_rng(5002)
    q8 = rng.standard_normal((8, 1024)).astype(np.float32); q8 /= np.linalg.norm(q8, axis=1, keepdims=True)
    bids = bidx.search(q8, 100)[0]
    us_bs = _median_us(lambda: bidx.search(q8, 100), 20)
    us_br = _median_us(lambda: bidx.rescore(q8, bids, 20), 20)
    scan_bytes = big * 1024 * 2 + 8 * 1024 * 4 + 8 * 100 * 12
    out["at_1M_rows_1024d_f16"] = {"search_B8_k100_us": us_bs, "rescore_100_to_20_us": us_br, "frac": scan_bytes / (us_bs * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "frac_of": "8 TB/s HBM, algorithmic bytes of the scan / the whole synchronous call"}
    out["frac"] = enc["frac"]
    out["frac_of"] = "the encode dominates this configuration: fraction of 2.5 PFLOP/s fp16 MFMA of the forward (see encode); the searches are latency-bound at 391 rows"
    bidx.close()
    if hasattr(em, "close"):
        em.close()
    del em
    try:        # CPU twin: the reference's batch_encode restated, same random-init BERT-large, fp32, a few chunks
        from oracle import encode_torch as enc_o
        from tools.synthetic import random_bert
        tok, _ = synthetic_wordpiece_tokenizer()
        model = random_bert("large", vocab_size=len(tok))
        enc_o.encode(model, tok, chunks[:1], instruction=enc_o.BGE_PREFIX)
        t0 = time.perf_counter(); enc_o.encode(model, tok, chunks[:cpu_chunks], instruction=enc_o.BGE_PREFIX); dt = time.perf_counter() - t0
        import torch as _t
        out["cpu"] = {"what": "oracle.encode_torch.encode = embedding_model/BGEEmbedding.py:92-129 on the CPU (fp32, same BERT-large shape)",
                      "chunks": cpu_chunks, "chunks_per_s": cpu_chunks / dt, "cores": _t.get_num_threads()}
    except Exception as e:
        out["cpu"] = {"error": repr(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------------ f1
def f1_selfjoin(torch, device, entities=200_000, dim=768, thr=0.8, batch=1024, cpu_entities=20_000):
    """the synonymy self-join of ComoRAG.add_synonymy_edges (ComoRAG.py:670-712) — every entity against every
    entity, neighbours with cosine >= 0.8 — by the threshold search (cmr_index_search_min_score) vs by materialising 2047
    neighbours per entity and cutting afterwards (what the reference asks retrieve_knn for), bf16 and fp32 storage.  Timed on
    the WHOLE join for the threshold path (device-resident queries, one stream, one sync), on 8 / 2 batches of 1024 queries
    scaled to the whole join for the host-block variant and for materialise + select; the neighbours >= 0.8 of a batch are
    compared between the paths."""
    from comorag_amd.index import DenseIndex
    g = torch.Generator(device=device); g.manual_seed(3)
    x = torch.randn((entities, dim), generator=g, device=device)
    dup = torch.randint(0, entities, (entities // 10,), generator=g, device=device)      # 10 % of the entities get a near-duplicate
    x[:entities // 10] = x[dup] + 0.15 * x[:entities // 10]
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    xh = x.cpu().numpy()
    out = {"entities": entities, "dim": dim, "threshold": thr, "query_batch": batch}
    flops = 2.0 * entities * entities * dim
    for dtype in ("bf16", "f32"):
        idx = DenseIndex(dim, dtype, device=device.index or 0, capacity_hint=entities)
        idx.append_dev(x); torch.cuda.synchronize(device)
        def run(fn, nb):
            fn(xh[:batch]); t0 = time.perf_counter()
            for b in range(nb):
                fn(xh[b * batch:(b + 1) * batch])
            return (time.perf_counter() - t0) / nb
        t_thr = run(lambda q: idx.search_min_score(q, 128, thr), 8)
        # the WHOLE join the way retrieval.retrieve_knn(min_score=) runs it: queries resident on the device (they are the index's own
        # rows), blocks of `query_batch_size` = 1000 queries (the reference's default, utils/embed_utils.py:8) enqueued in throughput mode
        # (cmr_index_search_min_score_pipelined: packing of block i + 1 and the merge of block i - 1 beside the scan of block i), one
        # synchronisation, one download.  Best of three (round 5 timed the one-stream entry point here: 86-108 ms at bf16).
        ids_t = torch.empty((entities, 128), dtype=torch.int64, device=device); sc_t = torch.empty((entities, 128), dtype=torch.float32, device=device)
        qblock = 1000
        t_join, t_one, t_down = None, None, None
        for rep in range(3):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter(); done = None
            for b0 in range(0, entities, qblock):
                b1 = min(b0 + qblock, entities)
                done = idx.search_min_score_pipelined(x[b0:b1], 128, thr, ids_t[b0:b1], sc_t[b0:b1])
            idx.sync(done)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            t_join = dt if t_join is None else min(t_join, dt)
            # the [entities, 128] int64 ids to the host (205 MB through torch's pageable copy: what retrieve_knn does next), timed apart:
            # the join is the device's work, the download is the link's
            t1 = time.perf_counter()
            ids_h = ids_t.cpu().numpy()
            dl = time.perf_counter() - t1
            t_down = dl if t_down is None else min(t_down, dl)
        passes = sum((min(b0 + qblock, entities) - b0 + 255) // 256 for b0 in range(0, entities, qblock))
        # the one-stream entry point on the same blocks (no overlap between a block's packing, scan and merge), for the fixed cost of a pass
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for b0 in range(0, entities, qblock):
            b1 = min(b0 + qblock, entities)
            idx.search_min_score_dev(x[b0:b1], 128, thr, ids_t[b0:b1], sc_t[b0:b1])
        torch.cuda.synchronize(device)
        t_one = time.perf_counter() - t0
        one_ids = ids_t.cpu().numpy()
        del sc_t
        t_mat = run(lambda q: idx.search(q, 2047, with_minmax=False), 2)
        a = idx.search_min_score(xh[:batch], 128, thr); b = idx.search(xh[:batch], 2047, with_minmax=False)
        same = all(np.array_equal(a[0][i][a[0][i] >= 0], b[0][i][b[1][i] >= thr][:128]) for i in range(batch))
        scale = entities / batch
        peak = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else F32_PEAK_TFLOPS
        same = same and bool(np.array_equal(ids_h[:batch], a[0])) and bool(np.array_equal(ids_h, one_ids))
        ideal_us = 2.0 * 256 * entities * dim / (peak * 1e12) * 1e6 if dtype == "bf16" else None
        out[dtype] = {"threshold_search_whole_join_s": t_join, "route": f"throughput mode, blocks of {qblock} queries (retrieve_knn's), best of 3",
                      "ids_download_s": t_down, "passes_of_up_to_256_queries": passes, "us_per_pass": t_join / passes * 1e6, "one_stream_whole_join_s": t_one, "one_stream_us_per_pass": t_one / passes * 1e6,
                      "us_per_pass_at_the_dtype_peak": ideal_us,
                      "threshold_search_host_blocks_s": t_thr * scale, "materialise_select_k2047_s": t_mat * scale,
                      "speedup": t_mat * scale / t_join,
                      "same_neighbours_above_threshold": bool(same), "TFLOPs": flops / t_join / 1e12, "frac": flops / t_join / 1e12 / peak,
                      # bytes each path writes per 1024-query batch, by construction: the threshold path writes candidate keys
                      # (8 B each, only scores >= thr) + [B,128] results; the other a [B, N] fp32 score block + [B,2047] results
                      "bytes_written_per_batch_threshold": int(batch * 128 * 12 + (a[0] >= 0).sum() * 8),
                      "bytes_written_per_batch_materialise": int(batch * entities * 4 + batch * 2047 * 12)}
        idx.close()
    out["frac"] = out["bf16"]["frac"]
    out["frac_of"] = "dense MFMA peak (2.5 PFLOP/s bf16, 157 TFLOP/s fp32): the join is a 200 K x 200 K x 768 GEMM with a threshold epilogue"
    try:        # CPU twin at entity scale: retrieve_knn's torch.mm + torch.topk blocks forced onto the CPU
        from oracle import retrieval_np as orc
        sub = xh[:cpu_entities]
        t0 = time.perf_counter(); orc.retrieve_knn_torch_cpu(sub[:2000], sub, k=2047); dt = time.perf_counter() - t0
        import torch as _t
        out["cpu"] = {"what": "oracle.retrieve_knn_torch_cpu = utils/embed_utils.py:8-97 on the CPU (k = 2047)", "entities": cpu_entities, "queries_timed": 2000,
                      "s_for_the_sample_join": dt * cpu_entities / 2000, "s_scaled_to_full_join": dt * (entities / 2000) * (entities / cpu_entities),
                      "scaled": "quadratic in the entity count", "cores": _t.get_num_threads()}
    except Exception as e:
        out["cpu"] = {"error": repr(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------------ f4
def _ppr_case(torch, device, n_pass, n_ent, dim, dtype, reps, seed):
    from comorag_amd.index import DenseIndex
    from comorag_amd.ppr import DeviceGraph, ppr_passage_scores
    rng = np.random.default_rng(seed)
    idx = DenseIndex(dim, dtype, device=device.index or 0, capacity_hint=n_pass)
    for blk in _unit_rows_dev(torch, n_pass, dim, device, seed + 1):
        idx.append_dev(blk)
    torch.cuda.synchronize(device)
    nv = n_ent + n_pass
    passage_vertex = (n_ent + np.arange(n_pass)).astype(np.int32)
    src = np.concatenate([rng.integers(0, n_ent, 3 * n_pass), rng.integers(0, n_ent, 2 * n_ent)]).astype(np.int32)   # 3 entities per passage + entity-entity edges
    dst = np.concatenate([np.repeat(passage_vertex, 3), rng.integers(0, n_ent, 2 * n_ent)]).astype(np.int32)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = rng.uniform(0.5, 1.5, len(src))
    g = DeviceGraph(nv, src, dst, w, device=device.index or 0); g.set_passage_vertices(passage_vertex)
    q = rng.standard_normal(dim).astype(np.float32); q /= np.linalg.norm(q)
    phrase = np.zeros(nv); phrase[rng.integers(0, n_ent, 6)] = rng.uniform(0.2, 1.0, 6)
    fused_us = _median_us(lambda: ppr_passage_scores(idx, g, q, phrase, 0.05), reps)
    fused = ppr_passage_scores(idx, g, q, phrase, 0.05)
    # the unfused route on the same device pieces: complete ranking to the host (12 N bytes), host min-max + scatter
    # (ComoRAG.py:1034-1045), then the PageRank alone on the device
    from comorag_amd import retrieval
    def unfused():
        ids, sc = retrieval.dense_passage_retrieval(idx, q[None, :])
        reset = phrase.copy()
        reset[passage_vertex[ids]] = retrieval.min_max_normalize(sc) * 0.05
        return g.ppr(reset)[passage_vertex]
    unfused_us = _median_us(unfused, max(3, reps // 3), warm=1)
    close = bool(np.allclose(unfused(), fused, atol=1e-9))
    res = {"passages": n_pass, "entities": n_ent, "edges": int(len(src)), "dim": dim, "dtype": dtype, "iterations": 43,
           "fused_us_per_query": fused_us, "unfused_us_per_query": unfused_us, "speedup": unfused_us / fused_us,
           "bytes_returned_fused": 8 * n_pass, "bytes_returned_unfused": 12 * n_pass + 8 * nv, "fused_equals_unfused": close}
    # HBM bytes the fused call moves, by construction: the scan + per iteration (CSR entries 12 B + x gather 8 B per entry, 3 vectors of 8 B per vertex)
    ne2 = 2 * len(src)
    algo = n_pass * dim * (2 if dtype != "f32" else 4) + 43 * (ne2 * 20 + nv * 24) + n_pass * 24
    res["frac"] = algo / (fused_us * 1e-6) / 1e9 / HBM_PEAK_GBS
    res["frac_of"] = "8 TB/s HBM over scan + 43 power-iteration steps (latency-bound at ComoRAG scale: ~90 launches of a few us)"
    idx.close(); g.close()
    return res, (nv, src, dst, w, phrase)


def f4_ppr(torch, device):
    """DPR-seeded personalised PageRank per query, fused on the device (cmr_index_ppr) vs complete ranking to
    the host + host scatter + device PageRank, at ComoRAG scale (5 K passages / 1.5 K entities, 768-d) and at 1 M passages."""
    small, gsmall = _ppr_case(torch, device, 5_000, 1_500, 768, "f32", 30, 7001)
    big, _ = _ppr_case(torch, device, 1_000_000, 200_000, 768, "bf16", 6, 7002)
    out = {"comorag_scale": small, "at_1M_passages": big, "frac": big["frac"], "frac_of": big["frac_of"]}
    try:        # CPU twin: the PageRank the reference delegates to igraph / prpack — here networkx (scipy power iteration) and a
                # scipy.sparse direct solve of the same system, ComoRAG scale
        import networkx as nx
        import scipy.sparse as sp
        nv, src, dst, w, phrase = gsmall
        G = nx.Graph(); G.add_nodes_from(range(nv))
        for u, v, x in zip(src.tolist(), dst.tolist(), w.tolist()):
            if G.has_edge(u, v): G[u][v]["weight"] += x
            else: G.add_edge(u, v, weight=x)
        reset = phrase.copy(); reset[1_500:] = np.random.default_rng(1).uniform(0, 0.05, nv - 1_500)
        pers = {i: float(reset[i]) for i in range(nv)}
        t0 = time.perf_counter(); nx.pagerank(G, alpha=0.5, personalization=pers, weight="weight", dangling=pers, tol=1e-12); t_nx = time.perf_counter() - t0
        W = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([src, dst]), np.concatenate([dst, src]))), shape=(nv, nv)).tocsr()
        s = np.asarray(W.sum(axis=1)).ravel(); r = reset / reset.sum()
        inv = np.where(s > 0, 1.0 / np.where(s > 0, s, 1), 0.0)
        t0 = time.perf_counter()
        y = r.copy()
        for _ in range(43):
            y = 0.5 * (W.T @ (y * inv) + y[s == 0].sum() * r) + 0.5 * r
        t_sp = time.perf_counter() - t0
        out["cpu"] = {"what": "personalised PageRank alone at ComoRAG scale (the reference calls igraph/prpack, absent here): networkx.pagerank and 43 scipy.sparse power-iteration steps",
                      "networkx_ms": t_nx * 1e3, "scipy_power_iteration_ms": t_sp * 1e3, "vertices": nv, "cores": 1}
    except Exception as e:
        out["cpu"] = {"error": repr(e)[:200]}
    return out
