"""HipBGEEmbeddingModel — drop-in for BGEEmbeddingModel (src/comorag/embedding_model/BGEEmbedding.py).

tokenise (HF tokenizers, host threads) → encoder forward (PyTorch-ROCm GEMMs; for 16-bit BERT encoders the attention and
the bias + residual + LayerNorm stages are HIP kernels, fused_bert.py) → fused masked mean-pool +
L2-normalise as ONE HIP kernel pair on the encoder's output tensor (`cmr_pool_l2norm`; replaces
`mean_pooling` :15-28 and `F.normalize` :126-127).  Call surface, argument handling and quirks follow
the reference:
  * `batch_encode` ALWAYS overwrites `instruction` with the fixed BGE prefix because callers never
    pass `is_query` (:150-155), so `instruction=` / `norm=` passed by ComoRAG are ignored — kept;
  * prefix + text are concatenated with no separator (:109);
  * `encode(list)` positional returns a torch tensor [n, D] (memory_utils.py:176,205,297 index it).
Fixed consciously: `max_length` is clamped to the model's `max_position_embeddings` (the reference
default 2048 overflows BERT's 512 positions), `embedding_model_dtype` is honoured, and
tokenisation of the next mini-batches (a pool of host threads) overlaps the forward of mini-batch i; with more than
one mini-batch the prompts are grouped by token count (`embedding_length_bucketing`, default on): a row's embedding
does not depend on its neighbours in the mini-batch beyond GEMM rounding, and short chunks stop paying for long ones.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from typing import List, Optional, Union

import numpy as np

from .. import _lib as L
from ..utils.config_utils import BaseConfig, cfg_get
from .base import BaseEmbeddingModel, EmbeddingConfig, make_cache_embed

BGE_PREFIX = "Generate a representation for this sentence to retrieve relevant articles:"
_TORCH_TO_CMR = {"torch.float32": L.CMR_F32, "torch.bfloat16": L.CMR_BF16, "torch.float16": L.CMR_F16}


def tokenize_batch(tokenizer, prompts: List[str], max_length: int):
    """`tokenizer(prompts, padding=True, truncation=True, max_length=..., return_tensors="pt")` of
    BGEEmbedding.py:112-117 with the same int64 tensors, minus transformers' pure-Python
    `flatten()` of every id list during tensor conversion (a third of the call at 32 x 512 tokens,
    all of it under the GIL — it competes with the thread that launches the encoder's kernels)."""
    import torch
    enc = tokenizer(prompts, padding=True, truncation=True, max_length=int(max_length), return_tensors=None)
    return {k: torch.from_numpy(np.asarray(v, dtype=np.int64)) for k, v in enc.items()}


def tokenize_ragged(tokenizer, prompts: List[str], max_length: int):
    """Token ids per prompt, truncated, NOT padded (for length-bucketed mini-batches)."""
    enc = tokenizer(prompts, padding=False, truncation=True, max_length=int(max_length), return_tensors=None)
    return enc["input_ids"]


def pad_batch(tokenizer, id_lists, multiple: int = 1, limit: Optional[int] = None):
    """The tensors `tokenizer(..., padding=True, return_tensors="pt")` would build for these id lists (BERT-style
    inputs: input_ids right-padded with pad_token_id, attention_mask, token_type_ids all zero for single segments);
    `multiple` > 1 rounds the padded width up to that multiple (never beyond `limit`)."""
    import torch
    n, width = len(id_lists), max(len(x) for x in id_lists)
    if multiple > 1:
        width = max(width, min(-(-width // multiple) * multiple, limit if limit is not None else 1 << 30))
    ids = np.full((n, width), tokenizer.pad_token_id or 0, dtype=np.int64)
    mask = np.zeros((n, width), dtype=np.int64)
    for r, x in enumerate(id_lists):
        ids[r, :len(x)] = x
        mask[r, :len(x)] = 1
    out = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    if "token_type_ids" in getattr(tokenizer, "model_input_names", ()):
        out["token_type_ids"] = torch.zeros((n, width), dtype=torch.int64)
    return out


def pool_l2norm(hidden, mask, normalize: bool = True):
    """hidden [b,l,d] (fp32/bf16/fp16, CUDA, contiguous), mask [b,l] int64 → torch fp32 [b,d] on the
    same device, on torch's current stream.  Raises without a GPU: there is no CPU fallback."""
    import torch
    if not hidden.is_cuda:
        raise RuntimeError("pool_l2norm needs CUDA tensors: comorag_amd has no CPU fallback")
    hidden = hidden.contiguous()
    mask = mask.to(device=hidden.device, dtype=torch.int64).contiguous()
    b, l, d = hidden.shape
    out = torch.empty((b, d), dtype=torch.float32, device=hidden.device)
    stream = torch.cuda.current_stream(hidden.device).cuda_stream
    L.check(L.lib().cmr_pool_l2norm(hidden.device.index or 0, C.c_void_p(hidden.data_ptr()), _TORCH_TO_CMR[str(hidden.dtype)],
                                    C.c_void_p(mask.data_ptr()), b, l, d, 1 if normalize else 0,
                                    C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    return out


class _EncodeReplica:
    """One data-parallel copy of the encoder for corpus encodes: the layer stack on `device` (the owner's own model for the first
    replica, the same weights again for logical replicas on the owner's device, a deep copy on any other GPU), a stream and a worker
    thread of its own.  Captured mini-batch graphs belong to the replica's FusedBertLayers."""

    def __init__(self, owner, device, first: bool):
        import copy
        import torch
        from . import fused_bert
        self.device = device
        if first:
            self.model, self.fused = owner.embedding_model, owner._fused
        else:
            same = device == owner.device
            self.model = owner.embedding_model if same else copy.deepcopy(owner.embedding_model).to(device).eval()
            with torch.cuda.device(device):
                self.fused = fused_bert.FusedBertLayers(self.model, graphs=owner._fused.graphs, gelu="epilogue" if owner._fused.gelu_path.startswith("hipblaslt") else "exact")
        self.stream = torch.cuda.Stream(device)
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"cmr-enc-{device.index}")
        self.first = first

    def close(self) -> None:
        self.pool.shutdown(wait=True)
        if not self.first:
            self.fused.release()


class HipBGEEmbeddingModel(BaseEmbeddingModel):
    # batch_encode overwrites whatever `instruction=` it is handed with the fixed BGE prefix unless the caller passes is_query (nobody does): the
    # reference's behaviour (BGEEmbedding.py:150-155), kept.  Callers that would encode one text under two instructions (hooks.get_query_embeddings)
    # may therefore encode it once.
    instruction_is_ignored = True

    def __init__(self, global_config: Optional[BaseConfig] = None, embedding_model_name: Optional[str] = None,
                 model=None, tokenizer=None) -> None:
        """`model` / `tokenizer` may be injected (tests use a seed-initialised BertModel and a synthetic
        WordPiece vocabulary: no BGE weights exist offline); otherwise HF `from_pretrained` as the
        reference (:51-52)."""
        super().__init__(global_config=global_config)
        if embedding_model_name is not None:
            self.embedding_model_name = embedding_model_name
        self._init_embedding_config()
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("HipBGEEmbeddingModel needs an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", int(cfg_get(self.global_config, "device", 0)))
        if tokenizer is None or model is None:
            from transformers import AutoModel, AutoTokenizer
            tokenizer = tokenizer or AutoTokenizer.from_pretrained(self.embedding_model_name)
            model = model or AutoModel.from_pretrained(self.embedding_model_name, trust_remote_code=True)
        dt = str(cfg_get(self.global_config, "embedding_model_dtype", "auto")).lower()
        tdt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16,
               "f16": torch.float16}.get(dt)
        self.tokenizer = tokenizer
        self.embedding_model = model.to(self.device) if tdt is None else model.to(self.device, dtype=tdt)
        self.embedding_model.eval()
        self.embedding_dim = self.embedding_model.config.hidden_size
        # 16-bit BERT encoders run their layers through fused_bert.FusedBertLayers (HIP attention and bias + residual +
        # LayerNorm stages around PyTorch's GEMMs); anything else keeps the transformers forward.  `encoder_path` names it.
        self._fused, self.encoder_path = None, "transformers"
        if bool(cfg_get(self.global_config, "embedding_fused_encoder", True)):
            from . import fused_bert
            reason = fused_bert.why_not(self.embedding_model)
            if reason is None and getattr(tokenizer, "padding_side", "right") != "right":
                reason = "left-padding tokenizer"
            if reason is None:
                self._fused = fused_bert.FusedBertLayers(self.embedding_model, graphs=int(cfg_get(self.global_config, "embedding_hip_graphs", 24)),
                                                         gelu=str(cfg_get(self.global_config, "embedding_gelu", "exact")))
                self.encoder_path = "hip-fused-layers"
            else:
                self.encoder_path = f"transformers ({reason})"
        self.max_positions = int(getattr(self.embedding_model.config, "max_position_embeddings", 1 << 30))
        if getattr(self.embedding_model.config, "model_type", "") in ("roberta", "xlm-roberta", "camembert"):
            # RoBERTa-family position ids start at padding_idx + 1: a table of 8194 rows serves sequences of 8192 tokens
            pad = getattr(getattr(self.embedding_model, "embeddings", None), "padding_idx", None)
            pad = getattr(self.embedding_model.config, "pad_token_id", 1) if pad is None else pad
            self.max_positions = max(1, self.max_positions - int(pad or 0) - 1)
        import os
        # mini-batches of similar token count (sorted by length, results scattered back): a mini-batch is padded to ITS
        # longest prompt, so mixing a 40-token and a 512-token chunk wastes 92 % of the short one's forward
        self._bucket = bool(cfg_get(self.global_config, "embedding_length_bucketing", True))
        self._tok_workers = max(1, min(int(cfg_get(self.global_config, "embedding_tokenizer_threads", 2)), os.cpu_count() or 1))
        self._tok_pool = ThreadPoolExecutor(max_workers=self._tok_workers, thread_name_prefix="cmr-tok")
        # optional tokenizer PROCESSES (the Rust tokenizer holds the GIL while it encodes: threads share the core that also
        # launches the encoder's kernels).  Spawned, tokenizers-only workers (_tokworker.py); used by the length-bucketed path.
        import threading
        self._fast_tok = hasattr(tokenizer, "backend_tokenizer") and hasattr(tokenizer.backend_tokenizer, "to_str")
        self._bt_copies, self._bt_lock = {}, threading.Lock()
        # 0 (default): threads only; -1: processes are started by the first corpus-sized batch_encode call (>= two bucketing windows of
        # texts), in the background — that call goes on with threads until the workers answer; N > 0: N processes from the start.
        # (Same-box probe of every mode at BERT-base bf16, tools/tok_mode_probe.py: end to end 0.78-0.92 of forward-only for ALL of them,
        # run-to-run spread larger than any difference between them — the lever was the tokenizer call itself, see _ragged.)
        self._tok_procs, self._tok_procs_starting, self._closed = None, None, False
        n_procs = int(cfg_get(self.global_config, "embedding_tokenizer_processes", 0))
        self._tok_procs_auto = (min(4 if n_procs == -1 else -n_procs, max(1, (os.cpu_count() or 2) // 4)) if n_procs < 0 else 0) if self._fast_tok else 0
        if n_procs > 0 and self._fast_tok:
            self._tok_procs = self._start_tok_procs(n_procs)
        # Corpus encode over the node's GPUs (the reference: `device_map="auto"  # Use multiple GPUs if available`, BGEEmbedding.py:77).
        # Chunks are independent: one replica of the fused layer stack per entry of `embedding_devices` (a device may repeat, and
        # `embedding_encode_replicas` > the device count goes round them: logical replicas — how a one-GPU box runs this path), each with its
        # own stream, captured graphs and worker thread; a corpus-sized batch_encode deals its bucketing windows round the replicas and
        # gathers the rows, device to device, in arrival order (`_run_window`).  Replica 0 is this object's own model.
        self._replicas = []
        devs = cfg_get(self.global_config, "embedding_devices", None)
        n_rep = int(cfg_get(self.global_config, "embedding_encode_replicas", 0) or 0)
        if (devs or n_rep > 1) and self._fused is not None:
            devs = [int(d) for d in (devs or [self.device.index or 0])]
            n_vis = torch.cuda.device_count()
            if any(d < 0 or d >= n_vis for d in devs):
                raise ValueError(f"embedding_devices {devs}: this process sees {n_vis} GPU(s)")
            n_rep = max(n_rep, len(devs))
            if n_rep > 1:
                for r in range(n_rep):
                    self._replicas.append(_EncodeReplica(self, torch.device("cuda", devs[r % len(devs)]), first=(r == 0)))
        import collections, threading
        self._qcache, self._qcache_lock = collections.OrderedDict(), threading.Lock()
        self._qcache_size = int(cfg_get(self.global_config, "embedding_query_cache", 256) or 0)
        self._cached = bool(cfg_get(self.global_config, "embedding_cache_enabled", False))
        if self._cached:
            path = cfg_get(self.global_config, "embedding_cache_path", None) or "bge_embeddings_cache.db"
            self.encode = make_cache_embed(self._encode, path, self.device)
        else:
            self.encode = self._encode

    def _start_tok_procs(self, n: int):
        """A pool of spawned, tokenizers-only worker processes (_tokworker.py), answering before it is handed out."""
        import multiprocessing as mp
        import os
        from . import _tokworker
        tok = self.tokenizer
        pool = mp.get_context("spawn").Pool(min(int(n), os.cpu_count() or 1), initializer=_tokworker.init,
                                            initargs=(tok.backend_tokenizer.to_str(), getattr(tok, "truncation_side", "right"), "longest_first"))
        try:        # the workers have imported `tokenizers` and rebuilt the tokenizer — or the pool goes (workers that die in their
                    # initializer are respawned by mp.Pool for ever: a blocking apply() would never return)
            pool.apply_async(_tokworker.ragged, (["warm up"], 16)).get(timeout=120.0)
        except BaseException:
            pool.terminate()
            raise
        return pool

    def _maybe_start_tok_procs(self, n_texts: int, window_texts: int) -> None:
        """embedding_tokenizer_processes = -1: a corpus-sized call (>= two windows) starts the worker processes on a background
        thread; whichever call finds them answering uses them from its next window on."""
        if self._tok_procs is not None or not self._tok_procs_auto or self._tok_procs_starting is not None or n_texts < 2 * window_texts:
            return
        import threading

        def _go():
            try:
                pool = self._start_tok_procs(self._tok_procs_auto)
            except Exception:                     # no worker processes on this host: the threads stay
                self._tok_procs_auto = 0
                return
            with self._bt_lock:                   # close() may have given up waiting for this thread: a pool published now would never be terminated
                if self._closed:
                    pool.terminate()
                else:
                    self._tok_procs = pool
        with self._bt_lock:                  # ComoRAG encodes from up to 16 threads over one model instance: ONE of them starts the workers
            if self._tok_procs_starting is not None:
                return
            t = threading.Thread(target=_go, name="cmr-tok-procs", daemon=True)
            t.start()
            self._tok_procs_starting = t

    def _init_embedding_config(self) -> None:
        self.embedding_config = EmbeddingConfig.from_dict({
            "embedding_model_name": self.embedding_model_name,
            "norm": cfg_get(self.global_config, "embedding_return_as_normalized", True),
            "model_init_params": {"pretrained_model_name_or_path": self.embedding_model_name, "trust_remote_code": True,
                                  "device_map": "auto"},
            "encode_params": {"max_length": cfg_get(self.global_config, "embedding_max_seq_len", 2048),
                              "query_instruction": BGE_PREFIX, "passage_instruction": BGE_PREFIX,
                              "batch_size": cfg_get(self.global_config, "embedding_batch_size", 32), "num_workers": 32},
        })

    # ------------------------------------------------------------------ one mini-batch
    def _backend(self, max_length: int):
        """A private copy of the fast tokenizer's Rust backend with truncation to `max_length` and no padding, made once per
        length and never changed afterwards.  `tokenizer(...)` re-configures truncation / padding on the SHARED backend on
        every call (two threads asking for different padding race), and with padding off its Python post-processing is
        a third of the backend's own speed (measured 0.87 K vs 3.3 K chunks/s for 512-token chunks on 8 cores)."""
        bt = self._bt_copies.get(int(max_length))
        if bt is None:
            with self._bt_lock:
                bt = self._bt_copies.get(int(max_length))
                if bt is None:
                    from tokenizers import Tokenizer
                    bt = Tokenizer.from_str(self.tokenizer.backend_tokenizer.to_str())
                    bt.enable_truncation(int(max_length), stride=0, strategy="longest_first", direction=getattr(self.tokenizer, "truncation_side", "right"))
                    bt.no_padding()
                    self._bt_copies[int(max_length)] = bt
        return bt

    def _ragged(self, prompts: List[str], max_length: int):
        """Token ids per prompt, truncated, not padded: what `tokenizer(prompts, truncation=True, max_length=...)` yields — as int32
        arrays (made on the tokenizer's thread: the launching thread only ever concatenates them)."""
        if self._fast_tok:
            # encode_batch_fast (tokenizers >= 0.20) skips the character-offset bookkeeping nothing here reads: the same ids at ~5x the rate
            # (measured 0.8-1.1 K -> 4.7-5.2 K chunks/s of 512 tokens on 8 cores), i.e. a tokenizer that stays ahead of a 16-bit forward
            bt = self._backend(max_length)
            enc = bt.encode_batch_fast(list(prompts)) if hasattr(bt, "encode_batch_fast") else bt.encode_batch(list(prompts))
            return [np.asarray(e.ids, dtype=np.int32) for e in enc]
        return [np.asarray(x, dtype=np.int32) for x in tokenize_ragged(self.tokenizer, prompts, max_length)]

    def _forward_ragged(self, id_arrays, normalize: bool, fused=None):
        """One mini-batch of the fused stack from ragged id arrays: ONE int32 array lens | offsets | ids goes to the device
        (fused_bert.FusedBertLayers.forward_ragged); returns the pooled rows [b, D] fp32 on the GPU (`fused`: a replica's stack)."""
        b = len(id_arrays)
        head = np.empty(2 * b + 1, dtype=np.int32)
        lens = head[:b]
        lens[:] = [len(x) for x in id_arrays]
        head[b] = 0
        np.cumsum(lens, out=head[b + 1:])
        width = -(-int(lens.max()) // 16) * 16        # rows are padded (on the device) to the longest one, rounded up to 16 tokens
        return (fused or self._fused).forward_ragged(np.concatenate([head, *id_arrays]), b, width, normalize)

    @staticmethod
    def _window_groups(lens, batch_size: int, budget: int):
        """Mini-batches of one bucketing window: stable sort by token count, then as many rows as fit the token budget of a full-length
        batch (batch_size x max_length padded tokens; at most 8 x batch_size rows) — at bf16 a 32-row forward of short chunks is launch-bound."""
        order = np.argsort(lens, kind="stable")
        groups, start = [], 0
        while start < len(order):
            end = start + 1
            while end < len(order) and end - start < 8 * batch_size and (end - start + 1) * lens[order[end]] <= budget:
                end += 1
            groups.append(order[start:end])
            start = end
        return groups

    def _run_window(self, rep, id_lists, normalize: bool, batch_size: int, budget: int):
        """One bucketing window on one replica (its worker thread): the window's mini-batches through the replica's layer stack on the
        replica's stream; the rows come back in the WINDOW's order, fp32 [n, D] on the replica's device, complete (stream drained)."""
        import torch
        lens = np.array([len(x) for x in id_lists])
        groups = self._window_groups(lens, batch_size, budget)
        with torch.cuda.device(rep.device), torch.cuda.stream(rep.stream):
            parts = [self._forward_ragged([id_lists[j] for j in g], normalize, rep.fused) for g in groups]
            rows = torch.empty((len(id_lists), parts[0].shape[1]), dtype=parts[0].dtype, device=rep.device)
            where = torch.from_numpy(np.concatenate(groups)).pin_memory().to(rep.device, non_blocking=True)
            rows[where] = torch.cat(parts, dim=0)
        rep.stream.synchronize()
        return rows

    def _ragged_ok(self) -> bool:
        """Can mini-batches go to the device as ragged ids (fused stack, right-padding single-segment tokenizer, every row non-empty)?"""
        return self._fused is not None and getattr(self.tokenizer, "padding_side", "right") == "right" and self._fused.can_pool(16)

    def _tokenize(self, prompts: List[str], max_length: int):
        """The tensors of `tokenizer(prompts, padding=True, truncation=True, max_length=..., return_tensors="pt")` (:112-117)."""
        ml = min(int(max_length), self.max_positions)
        if self._fast_tok and getattr(self.tokenizer, "padding_side", "right") == "right":
            # with the fused layer stack the width is rounded up to a multiple of 16 (masked columns change no real token's
            # row): short queries of 5 .. 40 tokens then share a handful of captured mini-batch shapes
            return pad_batch(self.tokenizer, self._ragged(prompts, ml), multiple=16 if getattr(self, "_fused", None) is not None else 1, limit=ml)
        return tokenize_batch(self.tokenizer, prompts, ml)

    def _forward_pool(self, inputs, normalize: bool):
        import torch
        with torch.no_grad():
            # pinned staging + non-blocking copies: the id tensors of mini-batch i+1 cross the link while batch i computes
            lens = None
            if self._fused is not None and not inputs["attention_mask"].is_cuda:
                from .fused_bert import lens_of_mask
                lens = lens_of_mask(inputs["attention_mask"].numpy())      # None unless every row is ones-then-zeros
            inputs = {k: (v if v.is_cuda else self._upload(v)) for k, v in inputs.items()}
            if lens is not None:
                # (16-token-aligned mini-batches — what _tokenize pads to for the fused stack — take the tail inside the last
                # layer's LayerNorm kernel; any other shape pools the stored hidden state)
                return self._fused(inputs["input_ids"], self._upload(torch.from_numpy(lens)), token_type_ids=inputs.get("token_type_ids"),
                                   consume=lambda hidden: pool_l2norm(hidden, inputs["attention_mask"], normalize=normalize),
                                   pool=bool(normalize) if self._fused.can_pool(int(inputs["input_ids"].shape[1])) else None)
            hidden = self.embedding_model(**inputs).last_hidden_state
            return pool_l2norm(hidden, inputs["attention_mask"], normalize=normalize)

    def _upload(self, host):
        """Host tensor → device, asynchronously, on the caller's stream (pinned: a copy from pageable memory makes the host wait
        for everything queued on the stream).  Measured and dropped (profiles/r3_measurements.md): a staging ring of this
        model's own — every block stays busy until the GPU has worked off the forwards queued before its copy, so the host
        ends up waiting on the oldest; the same ring on a copy stream of its own moves that wait into the device allocator."""
        return host.pin_memory().to(self.device, non_blocking=True)

    def _encode(self, prompts: Union[str, List[str]], **kwargs):
        """BGEEmbedding.py:92-129 for one mini-batch; returns a torch fp32 tensor [b, D] on the GPU."""
        if isinstance(prompts, str):
            prompts = [prompts]
        instruction = kwargs.get("instruction", "")
        if instruction:
            prompts = [instruction + text for text in prompts]
        max_length = kwargs.get("max_length", self.embedding_config.encode_params.get("max_length", 512))
        if self._ragged_ok():
            ids = self._ragged(prompts, min(int(max_length), self.max_positions))
            if all(len(x) for x in ids):
                return self._forward_ragged(ids, kwargs.get("normalize", True))
        return self._forward_pool(self._tokenize(prompts, max_length), kwargs.get("normalize", True))

    # ------------------------------------------------------------------ public
    def batch_encode(self, texts: Union[str, List[str]], **kwargs) -> np.ndarray:
        """BGEEmbedding.py:131-185.  Returns np.ndarray [n, D] fp32, rows L2-normalised."""
        import torch
        if isinstance(texts, str):
            texts = [texts]
        return_device = bool(kwargs.pop("_return_device", False))
        params = deepcopy(self.embedding_config.encode_params)
        if kwargs:
            params.update(kwargs)
        if "is_query" in kwargs and kwargs["is_query"]:
            params["instruction"] = params.get("query_instruction", BGE_PREFIX)
        else:
            params["instruction"] = params.get("passage_instruction", BGE_PREFIX)
        batch_size = params.pop("batch_size", 16)
        # One string — a question: ComoRAG encodes the same question three times per tri_retrieve (once per query instruction, ComoRAG.py:921-935,
        # and once more for the episodic layer, utils/embed_utils.py get_similar_summaries), and every one of those is the same prompt through the
        # same deterministic forward (0.75 ms at BGE-base).  The last `embedding_query_cache` single-string results are kept by (prompt, max_length,
        # normalisation) and handed back as copies; 0 switches it off.  Corpus-sized calls never look here.
        qkey = None
        if len(texts) == 1 and not return_device and self._qcache_size > 0 and isinstance(texts[0], str):
            qkey = (params.get("instruction", "") + texts[0], int(params.get("max_length", 512)), bool(params.get("normalize", True)),
                    bool(self.embedding_config.norm and not kwargs.get("normalize", True)))
            with self._qcache_lock:
                hit = self._qcache.get(qkey)
                if hit is not None:
                    self._qcache.move_to_end(qkey)
                    return hit.copy()
        if len(texts) <= batch_size or self._cached:      # (not `self.encode is not self._encode`: bound methods are never identical)
            if len(texts) <= batch_size:
                params["prompts"] = texts
                results = self.encode(**params)
            else:  # cached encoder: keep the reference's simple loop
                parts = []
                for i in range(0, len(texts), batch_size):
                    params["prompts"] = texts[i:i + batch_size]
                    parts.append(self.encode(**params))
                results = torch.cat(parts, dim=0)
        else:
            # same mini-batches as the reference loop (:168-175); the next mini-batches are tokenised on host
            # threads while batch i is on the GPU.  tokenizers 0.22 holds the GIL in encode_batch, so more
            # than a couple of workers buys nothing (measured 1 vs 8 threads: 4134 vs 4092 chunks/s)
            instr = params.get("instruction", "")
            max_length = params.get("max_length", 512)
            normalize = params.get("normalize", True)
            chunks = [texts[i:i + batch_size] for i in range(0, len(texts), batch_size)]
            ahead = self._tok_workers + 1
            if self._bucket:
                # Windows of `embedding_bucket_window` reference chunks (default 4: 128 texts at batch 32).  Per window:
                # 1. token ids (worker threads or processes), 2. stable sort by token count, 3. mini-batches under the
                # token budget of one full-length batch, each padded to its own longest, 4. forward + pool, 5. scatter
                # back.  The ids of the NEXT windows are tokenised while this window's mini-batches are on the GPU — with
                # one global sort (round 2) every text was tokenised before the first forward started: end to end was
                # tokenizer time PLUS forward time (3.4 K chunks/s against a 5.9 K forward-only rate at 512 tokens).
                ml = min(int(max_length), self.max_positions)
                win = max(1, int(cfg_get(self.global_config, "embedding_bucket_window", 4)))
                # (the first two windows are ONE chunk each: the first forward starts after 1 / win of a window's tokenizer time)
                sizes, left = [], len(chunks)
                while left > 0:
                    sizes.append(min(left, 1 if len(sizes) < 2 else win))
                    left -= sizes[-1]
                starts = np.cumsum([0] + sizes[:-1])
                windows = [chunks[a:a + n] for a, n in zip(starts, sizes)]
                from . import _tokworker
                self._maybe_start_tok_procs(len(texts), win * batch_size)
                rag = lambda c: self._ragged([instr + t for t in c] if instr else list(c), ml)

                def submit(w):      # worker processes as soon as they answer (the Rust tokenizer holds the GIL: threads share the launching core)
                    procs = self._tok_procs
                    if procs is not None:
                        return [procs.apply_async(_tokworker.ragged, ([instr + t for t in c] if instr else list(c), ml)) for c in w]
                    return [self._tok_pool.submit(rag, c) for c in w]
                collect = lambda jobs: [x for j in jobs for x in (j.get() if hasattr(j, "get") else j.result())]
                look = 3                                   # windows being tokenised ahead of the one on the GPU
                pending = [submit(w) for w in windows[:look]]
                # token budget of a mini-batch: `embedding_forward_batches` reference batches' worth.  The GEMMs of a 128 x 512-token
                # forward run 16 % faster per chunk than those of a 32 x 512 one (forward alone 7.4 -> 8.6 K chunks/s), but end to
                # end the larger staging copies cost the host more than that (6.5-7.1 K -> 2.7-3.4 K chunks/s): default 1
                fwb = max(1, int(cfg_get(self.global_config, "embedding_forward_batches", 1)))
                results, base, budget = None, 0, fwb * batch_size * ml
                trace = getattr(self, "_trace", None)       # a list: per window (seconds waiting for token ids, seconds launching)
                import time as _time
                # replicas (embedding_devices / embedding_encode_replicas): only for corpus-sized calls on the ragged fused path
                reps = self._replicas if (len(self._replicas) > 1 and len(windows) >= 2 and self._ragged_ok()) else None
                rep_jobs = []
                for wi in range(len(windows)):
                    t_w0 = _time.perf_counter()
                    id_lists = collect(pending[wi])
                    t_w1 = _time.perf_counter()
                    pending[wi] = None
                    if wi + look < len(windows):
                        pending.append(submit(windows[wi + look]))
                    lens = np.array([len(x) for x in id_lists])
                    if reps is not None and lens.min() > 0:
                        # the window goes to the next replica's worker thread; its rows are collected below, in arrival order
                        rep = reps[wi % len(reps)]
                        rep_jobs.append((base, len(id_lists), rep.pool.submit(self._run_window, rep, id_lists, normalize, batch_size, budget)))
                        base += len(id_lists)
                        if trace is not None:
                            trace.append((t_w1 - t_w0, _time.perf_counter() - t_w1))
                        continue
                    groups = self._window_groups(lens, batch_size, budget)
                    if self._ragged_ok() and lens.min() > 0:
                        parts = [self._forward_ragged([id_lists[j] for j in g], normalize) for g in groups]
                    else:
                        parts = [self._forward_pool(pad_batch(self.tokenizer, [id_lists[j] for j in g]), normalize) for g in groups]
                    if results is None:
                        results = torch.empty((len(texts), parts[0].shape[1]), dtype=parts[0].dtype, device=parts[0].device)
                    # (pinned index: a pageable copy would hold the host until this window's forwards have all finished)
                    where = self._upload(torch.from_numpy(np.concatenate(groups) + base))
                    results[where] = torch.cat(parts, dim=0)
                    base += len(id_lists)
                    if trace is not None:
                        trace.append((t_w1 - t_w0, _time.perf_counter() - t_w1))
                for at, n_w, job in rep_jobs:               # device to device (peer copy across GPUs), order restored by the window's position
                    rows = job.result()
                    if results is None:
                        results = torch.empty((len(texts), rows.shape[1]), dtype=rows.dtype, device=self.device)
                    results[at:at + n_w] = rows if rows.device == results.device else rows.to(results.device)
            else:
                prep = lambda c: self._tokenize([instr + t for t in c] if instr else list(c), max_length)
                futs = [self._tok_pool.submit(prep, c) for c in chunks[:ahead]]
                parts = []
                for i in range(len(chunks)):
                    inputs = futs[i].result()
                    futs[i] = None
                    if i + ahead < len(chunks):
                        futs.append(self._tok_pool.submit(prep, chunks[i + ahead]))
                    parts.append(self._forward_pool(inputs, normalize))
                results = torch.cat(parts, dim=0)
        if return_device and isinstance(results, torch.Tensor) and not (self.embedding_config.norm and not kwargs.get("normalize", True)):
            return results.float().contiguous()
        if isinstance(results, torch.Tensor):
            results = results.float().cpu().numpy()
        if self.embedding_config.norm and not kwargs.get("normalize", True):
            results = (results.T / np.linalg.norm(results, axis=1)).T
        if qkey is not None and isinstance(results, np.ndarray):
            with self._qcache_lock:
                self._qcache[qkey] = results.copy()
                while len(self._qcache) > self._qcache_size:
                    self._qcache.popitem(last=False)
        return results

    def batch_encode_dev(self, texts: Union[str, List[str]], **kwargs):
        """`batch_encode` that leaves the rows where the encoder put them: a torch fp32 CUDA tensor [n, D].
        EmbeddingStore.insert_strings appends it to the HBM index as it is (cmr_index_append_dev) instead of copying the
        rows to the host and uploading them again; the host copy the store keeps is made from the same tensor."""
        return self.batch_encode(texts, _return_device=True, **kwargs)

    def close(self) -> None:
        for rep in getattr(self, "_replicas", []):
            rep.close()
        self._replicas = []
        if getattr(self, "_fused", None) is not None:
            self._fused.release()
        starting = getattr(self, "_tok_procs_starting", None)
        if starting is not None and starting.is_alive():
            starting.join(30.0)                   # a pool still being spawned: let it finish, then take it down with the rest
        self._tok_procs_auto = 0
        lock = getattr(self, "_bt_lock", None)
        if lock is not None:
            with lock:                            # a starter thread that outlives the join sees the flag and terminates its own pool
                self._closed = True
                pool, self._tok_procs = getattr(self, "_tok_procs", None), None
            if pool is not None:
                pool.terminate()
        if getattr(self, "_tok_pool", None) is not None:
            self._tok_pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_queries(self, queries, **kwargs) -> np.ndarray:
        kwargs["is_query"] = True
        return self.batch_encode(queries, **kwargs)

    def encode_passages(self, passages, **kwargs) -> np.ndarray:
        kwargs["is_query"] = False
        return self.batch_encode(passages, **kwargs)
