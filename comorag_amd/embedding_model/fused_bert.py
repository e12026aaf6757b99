"""The encoder stack behind `self.embedding_model(**inputs)` (embedding_model/BGEEmbedding.py:119) for 16-bit BERT encoders.

transformers' BertLayer is ~20 kernels and as many Python module calls per layer: three projection GEMMs, SDPA with an additive
mask, the output GEMM, bias / residual adds, LayerNorm, the FFN GEMMs and GELU.  Here a layer is

    qkv  = x @ [Wq | Wk | Wv]^T + b          one hipBLASLt GEMM instead of three (PyTorch-ROCm, as north_star prescribes)
    ctx  = cmr_encoder_attention(qkv, lens)   HIP: masked softmax(QK^T/8)V straight off the packed projection, no head transposes
    x    = cmr_encoder_add_layernorm(ctx @ Wo^T, bo, x)       HIP: dense bias + residual + LayerNorm in one pass
    x    = cmr_encoder_add_layernorm(gelu(x @ W1^T + b1) @ W2^T, b2, x)      (bias in the up-projection GEMM's epilogue; GELU: the erf-form kernel)

seven host calls per layer — and ONE per forward once a mini-batch shape has been captured as a hipGraph (`graphs`) — so the
thread that launches the forward leaves the interpreter lock to the tokenizer threads.
The weights are the loaded model's own tensors (query / key / value concatenated once); the embedding gathers, their sum and
LayerNorm are one HIP kernel (cmr_encoder_embed_layernorm).  Results equal the transformers forward up to 16-bit rounding (the fused LayerNorm rounds once instead of
three times): tests/test_encoder_fused_gpu.py compares both with the fp32 oracle.

Used when `why_not(model)` is None: BERT / RoBERTa / XLM-R architecture (bge-base / -large, bge-m3), absolute positions, exact GELU,
64-wide heads, bf16 / fp16 weights; any other model keeps the transformers forward (`HipBGEEmbeddingModel.encoder_path` says which one runs).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib as L


# Graph captures are exclusive in the process — against each other AND against every other thread's forward launches:
#   * `torch.cuda.graph.__enter__` synchronises the whole DEVICE, which HIP refuses while another stream of it is capturing (two encoder
#     replicas — bge._EncodeReplica: one FusedBertLayers each, own stream and worker thread — capturing their first shapes at the same
#     moment: hipErrorStreamCaptureUnsupported);
#   * a forward launched EAGERLY by another thread meanwhile may be hipBLASLt's first call for its shape, and that call touches the
#     legacy stream: "operation would make the legacy stream depend on a capturing blocking stream" (hipblaslt.cpp:172), the GEMM fails,
#     the process dies in Tensile's initialisation (seen in bench.py --single-process: four replicas behind an index's blocking streams).
# Forwards hold the gate shared while they ENQUEUE (host side only: the GPU work of other streams runs on during a capture).
import threading as _threading


class _Gate:
    def __init__(self):
        self._cv, self._readers, self._writer = _threading.Condition(), 0, False

    class _Hold:
        def __init__(self, gate, exclusive): self.g, self.x = gate, exclusive
        def __enter__(self):
            g = self.g
            with g._cv:
                if self.x:
                    while g._writer or g._readers: g._cv.wait()
                    g._writer = True
                else:
                    while g._writer: g._cv.wait()
                    g._readers += 1
        def __exit__(self, *exc):
            g = self.g
            with g._cv:
                if self.x: g._writer = False
                else: g._readers -= 1
                g._cv.notify_all()

    def shared(self): return _Gate._Hold(self, False)
    def exclusive(self): return _Gate._Hold(self, True)


_GATE = _Gate()


# BERT and its RoBERTa-family twins (same layer; position ids start at padding_idx + 1, one token type): XLM-R is bge-m3, the one
# BGE model whose 8192 positions are safe with the reference's default embedding_max_seq_len = 2048 (config_utils.py:140)
ENCODER_TYPES = ("bert", "roberta", "xlm-roberta")


def position_offset(model) -> int:
    """First position id of a sequence: 0 for BERT, padding_idx + 1 for the RoBERTa family
    (transformers' RobertaEmbeddings.create_position_ids_from_input_ids)."""
    if getattr(model.config, "model_type", "") == "bert":
        return 0
    pad = getattr(model.embeddings, "padding_idx", None)
    if pad is None:
        pad = getattr(model.config, "pad_token_id", 1)
    return int(pad) + 1


def why_not(model) -> Optional[str]:
    """None if the fused layer stack can run this model, else the reason it cannot."""
    import torch
    cfg = getattr(model, "config", None)
    if cfg is None or getattr(cfg, "model_type", "") not in ENCODER_TYPES:
        return "not a BERT / RoBERTa / XLM-R encoder"
    if not (hasattr(model, "embeddings") and hasattr(model, "encoder") and hasattr(model.encoder, "layer")
            and all(hasattr(model.embeddings, n) for n in ("word_embeddings", "position_embeddings", "token_type_embeddings", "LayerNorm"))):
        return "unexpected module layout"
    if getattr(cfg, "position_embedding_type", None) not in (None, "absolute"):
        return "relative position embeddings"
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        return f"activation {cfg.hidden_act!r}"
    if getattr(cfg, "is_decoder", False) or getattr(cfg, "add_cross_attention", False) or getattr(cfg, "chunk_size_feed_forward", 0):
        return "decoder / cross-attention / chunked feed-forward configuration"
    if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads != 64:
        return "head width is not 64"
    if cfg.hidden_size % 4 or cfg.hidden_size > 2048 or cfg.intermediate_size % 4:
        return "hidden size not supported by the LayerNorm kernel"
    dt = next(model.parameters()).dtype
    if dt not in (torch.bfloat16, torch.float16):
        return f"{dt} weights (the fused layers are 16-bit)"
    return None


def lens_of_mask(mask: np.ndarray) -> Optional[np.ndarray]:
    """Token counts of a right-padded attention mask [b, l] (ones then zeros in every row), or None if it is anything else."""
    m = np.asarray(mask)
    if m.ndim != 2 or m.shape[1] == 0 or not np.all((m == 0) | (m == 1)):
        return None
    if m.shape[1] > 1 and np.any(m[:, 1:] > m[:, :-1]):
        return None
    lens = m.sum(axis=1).astype(np.int32)
    return lens if np.all(lens > 0) else None


def gelu_epilogue_available(device, dtype) -> bool:
    """Does this PyTorch-ROCm build run `torch._addmm_activation(bias, x, w.T, use_gelu=True)` as ONE hipBLASLt GEMM whose epilogue
    adds the bias and applies GELU in its tanh form?  Checked on the device, two ways: (i) the call agrees with tanh-GELU of the
    same linear map to within a 16-bit rounding step and NOT better with the erf form; (ii) torch's profiler sees ONE device kernel
    for the call — PyTorch's fallback (`addmm` + a separate tanh-GELU kernel) computes the same values in two launches: there the
    approximation would be paid for with no launch saved, and the exact path stays."""
    import torch
    import torch.nn.functional as F
    if not hasattr(torch, "_addmm_activation"):
        return False
    try:
        g = torch.Generator(device="cpu").manual_seed(5)
        x = (torch.randn((256, 256), generator=g) * 1.5).to(device=device, dtype=dtype)
        w = (torch.randn((512, 256), generator=g) / 16.0).to(device=device, dtype=dtype)
        b = torch.randn((512,), generator=g).to(device=device, dtype=dtype)
        got = torch._addmm_activation(b, x, w.t(), use_gelu=True).float()
        lin = F.linear(x.float(), w.float(), b.float())
        want_tanh, want_erf = F.gelu(lin, approximate="tanh"), F.gelu(lin)
        step = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        tol = step * (1.0 + want_tanh.abs())
        ok = bool(((got - want_tanh).abs() <= tol).all())
        if not (ok and float((got - want_tanh).abs().mean()) <= float((got - want_erf).abs().mean())):
            return False
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize(device)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            torch._addmm_activation(b, x, w.t(), use_gelu=True)
            torch.cuda.synchronize(device)
        kernels = [e for e in prof.key_averages() if getattr(e, "device_type", None) is not None and "DeviceType.CUDA" in str(e.device_type)
                   and "memcpy" not in e.key.lower() and "memset" not in e.key.lower()]
        return sum(e.count for e in kernels) == 1
    except Exception:
        return False


class FusedBertLayers:
    def __init__(self, model, graphs: int = 0, gelu: str = "exact"):
        """gelu = "exact" (default): FFN-up GEMM + bias in hipBLASLt, then PyTorch's erf-form GELU kernel — the function the model
        was trained with and the reference runs (BGEEmbedding.py:119-120, `hidden_act = "gelu"`).
        gelu = "epilogue" (opt-in, `embedding_gelu`): projection, bias and GELU are ONE hipBLASLt GEMM (`torch._addmm_activation`) —
        one read + write of the [b*l, 4*hidden] activation less per layer (~7 % of a BERT-base forward), but hipBLASLt's epilogue is
        the TANH form: a different function, |tanh form - erf form| <= 4.8e-4 per activation (largest near |x| ~ 2-3, where the value
        itself is ~1e-2: for small outputs the difference exceeds the 16-bit rounding step — bf16 ulp at 0.009 is 6e-5, fp16's 8e-6).
        Measured inside north_star's 1e-3 cosine bar on seed-initialised BERT shapes and on heavy-tailed pre-activations
        (tests/test_encoder_fused_gpu.py), not on real BGE weights (absent from the image) — hence not the default.  Falls back to
        "exact" when the build does not fuse it into a single launch (`gelu_path` says which one runs)."""
        import torch
        reason = why_not(model)
        if reason is not None:
            raise ValueError("FusedBertLayers: " + reason)
        if gelu not in ("epilogue", "exact"):
            raise ValueError("FusedBertLayers: gelu must be 'epilogue' or 'exact'")
        self.model = model
        cfg = model.config
        self.hidden, self.n_heads, self.eps = int(cfg.hidden_size), int(cfg.num_attention_heads), float(cfg.layer_norm_eps)
        self.dtype = next(model.parameters()).dtype
        self.cmr_dtype = L.CMR_BF16 if self.dtype == torch.bfloat16 else L.CMR_F16
        self.device = next(model.parameters()).device
        emb = model.embeddings
        self.emb = tuple(t.detach().contiguous() for t in (emb.word_embeddings.weight, emb.position_embeddings.weight,
                                                           emb.token_type_embeddings.weight, emb.LayerNorm.weight, emb.LayerNorm.bias))
        self.pos_offset = position_offset(model)
        self.gelu_path = "hipblaslt-epilogue-tanh" if (gelu == "epilogue" and gelu_epilogue_available(self.device, self.dtype)) else "exact-erf-kernel"
        import threading
        self.fold_pool = True                        # the encoder tail (mean-pool + L2-norm) rides the last layer's LayerNorm kernel
        self.graphs = int(graphs)                    # mini-batch shapes kept as captured hipGraphs (0: every forward is launched eagerly)
        self._graphs, self._seen, self._glock = {}, {}, threading.Lock()
        self.layers = []
        with torch.no_grad():
            for lyr in model.encoder.layer:
                att, so = lyr.attention.self, lyr.attention.output
                wqkv = torch.cat([att.query.weight, att.key.weight, att.value.weight], dim=0).contiguous()
                bqkv = torch.cat([att.query.bias, att.key.bias, att.value.bias], dim=0).contiguous()
                self.layers.append(tuple(t.detach().contiguous() for t in (
                    wqkv, bqkv, so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias,
                    lyr.intermediate.dense.weight, lyr.intermediate.dense.bias, lyr.output.dense.weight, lyr.output.dense.bias,
                    lyr.output.LayerNorm.weight, lyr.output.LayerNorm.bias)))

    # ------------------------------------------------------------------ the two HIP stages (also called by the tests)
    def attention(self, qkv, lens_dev, b: int, l: int, stream: Optional[int] = None):
        import torch
        out = torch.empty((b * l, self.hidden), dtype=qkv.dtype, device=qkv.device)
        if stream is None:
            stream = torch.cuda.current_stream(qkv.device).cuda_stream
        L.check(L.lib().cmr_encoder_attention(qkv.device.index or 0, C.c_void_p(qkv.data_ptr()), self.cmr_dtype, C.c_void_p(lens_dev.data_ptr()),
                                              b, l, self.n_heads, 64, C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def embed(self, input_ids, token_type_ids=None, stream: Optional[int] = None):
        """BertEmbeddings of a [b, l] id tensor → [b*l, hidden]."""
        import torch
        b, l = input_ids.shape
        word, pos, typ, gamma, beta = self.emb
        ids = input_ids.contiguous()
        tt = token_type_ids.contiguous() if token_type_ids is not None else None
        out = torch.empty((b * l, self.hidden), dtype=word.dtype, device=ids.device)
        if stream is None:
            stream = torch.cuda.current_stream(ids.device).cuda_stream
        L.check(L.lib().cmr_encoder_embed_layernorm(ids.device.index or 0, C.c_void_p(ids.data_ptr()), C.c_void_p(tt.data_ptr() if tt is not None else 0),
                                                    C.c_void_p(word.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(typ.data_ptr()),
                                                    C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), self.eps, b * l, l, self.hidden,
                                                    word.shape[0], pos.shape[0], typ.shape[0], int(getattr(self, "pos_offset", 0)), self.cmr_dtype,
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def embed_ragged(self, ids32, offsets, b: int, l: int, stream: Optional[int] = None):
        """BertEmbeddings straight from ragged token ids (int32 CUDA tensors: the b sequences back to back + their b + 1 starts)
        → [b*l, hidden]; rows behind a sequence's end are padding."""
        import torch
        word, pos, typ, gamma, beta = self.emb
        out = torch.empty((b * l, self.hidden), dtype=word.dtype, device=ids32.device)
        if stream is None:
            stream = torch.cuda.current_stream(ids32.device).cuda_stream
        L.check(L.lib().cmr_encoder_embed_layernorm_ragged(ids32.device.index or 0, C.c_void_p(ids32.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                                           C.c_void_p(word.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(typ.data_ptr()),
                                                           C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), self.eps, b, l, self.hidden,
                                                           word.shape[0], pos.shape[0], int(getattr(self, "pos_offset", 0)), self.cmr_dtype,
                                                           C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def add_layernorm(self, y, bias, residual, gamma, beta, stream: Optional[int] = None):
        import torch
        out = torch.empty_like(y)
        if stream is None:
            stream = torch.cuda.current_stream(y.device).cuda_stream
        L.check(L.lib().cmr_encoder_add_layernorm(y.device.index or 0, C.c_void_p(y.data_ptr()), C.c_void_p(bias.data_ptr() if bias is not None else 0),
                                                  C.c_void_p(residual.data_ptr() if residual is not None else 0), C.c_void_p(gamma.data_ptr()),
                                                  C.c_void_p(beta.data_ptr()), self.eps, y.shape[0], y.shape[1], self.cmr_dtype,
                                                  C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def add_layernorm_pool(self, y, bias, residual, gamma, beta, lens_dev, b: int, l: int, normalize: bool = True, stream: Optional[int] = None):
        """The last layer's add_layernorm with mean_pooling + F.normalize folded in (cmr_encoder_add_layernorm_pool): [b, hidden]
        fp32; the [b, l, hidden] state is never written."""
        import torch
        out = torch.empty((b, self.hidden), dtype=torch.float32, device=y.device)
        partial = torch.empty((b * (l // 16), self.hidden), dtype=torch.float32, device=y.device)
        if stream is None:
            stream = torch.cuda.current_stream(y.device).cuda_stream
        L.check(L.lib().cmr_encoder_add_layernorm_pool(y.device.index or 0, C.c_void_p(y.data_ptr()), C.c_void_p(bias.data_ptr() if bias is not None else 0),
                                                       C.c_void_p(residual.data_ptr() if residual is not None else 0), C.c_void_p(gamma.data_ptr()),
                                                       C.c_void_p(beta.data_ptr()), self.eps, b, l, self.hidden, self.cmr_dtype, C.c_void_p(lens_dev.data_ptr()),
                                                       1 if normalize else 0, C.c_void_p(partial.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
        return out

    def can_pool(self, l: int) -> bool:
        """Can the encoder tail be folded into the last layer's LayerNorm for mini-batches of l (padded) tokens?"""
        return self.fold_pool and l % 16 == 0 and self.hidden % 8 == 0

    # ------------------------------------------------------------------ forward
    def _stack(self, input_ids, lens_dev, token_type_ids, pool=None):
        """Embeddings + every layer on torch's current stream; all arguments on the GPU.  pool = None: the last hidden state
        [b, l, hidden]; pool = True / False: the pooled rows [b, hidden] fp32, L2-normalised or not (the tail rides the last
        layer's LayerNorm kernel)."""
        import torch
        import torch.nn.functional as F
        b, l = input_ids.shape
        stream = torch.cuda.current_stream(input_ids.device).cuda_stream
        return self._layers(self.embed(input_ids, token_type_ids, stream), lens_dev, b, l, pool, stream)

    def _stack_ragged(self, packed, b: int, l: int, pool):
        """The same from ONE int32 tensor lens[b] | offsets[b + 1] | token ids (ragged, back to back)."""
        import torch
        stream = torch.cuda.current_stream(packed.device).cuda_stream
        x = self.embed_ragged(packed[2 * b + 1:], packed[b:2 * b + 1], b, l, stream)
        return self._layers(x, packed[:b], b, l, pool, stream)

    def _layers(self, x, lens_dev, b: int, l: int, pool, stream):
        import torch
        import torch.nn.functional as F
        last = len(self.layers) - 1
        for n, (wqkv, bqkv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2) in enumerate(self.layers):
            qkv = F.linear(x, wqkv, bqkv)
            ctx = self.attention(qkv, lens_dev, b, l, stream)
            x = self.add_layernorm(F.linear(ctx, wo), bo, x, g1, be1, stream)
            if self.gelu_path == "hipblaslt-epilogue-tanh":
                h = torch._addmm_activation(b1, x, w1.t(), use_gelu=True)       # one GEMM: + bias, GELU in the epilogue
            else:
                h = F.gelu(F.linear(x, w1, b1))
            if n == last and pool is not None:
                return self.add_layernorm_pool(F.linear(h, w2), b2, x, g2, be2, lens_dev, b, l, bool(pool), stream)
            x = self.add_layernorm(F.linear(h, w2), b2, x, g2, be2, stream)
        return x.view(b, l, self.hidden)

    def __call__(self, input_ids, lens: np.ndarray, token_type_ids=None, consume=None, pool=None):
        """input_ids [b, l] int64 on the GPU (right-padded), lens[b] real token counts (host array, or an int32 tensor already
        on the GPU) → last hidden state [b, l, hidden];
        with `consume`, returns consume(hidden) instead; with pool = True / False (and `can_pool(l)`), the pooled rows [b, hidden]
        fp32 — masked mean over the tokens, L2-normalised or not — computed inside the last layer's LayerNorm kernel.

        A mini-batch shape seen before runs as ONE captured hipGraph (`graphs` > 0: up to that many shapes are kept, captured at
        a shape's second occurrence): ~100 launches and as many interpreter round trips become one, which is most of a short
        query's encode time and leaves the interpreter lock to the tokenizer threads during corpus encodes.  A graph's inputs and
        output are its own static buffers, so replays are serialised by a lock and `consume` (the pooling kernel's launch) runs
        under it: the next replay is stream-ordered behind the reader of this one's output."""
        import torch
        b, l = input_ids.shape
        with torch.no_grad():
            # (pinned: a copy from pageable memory makes the host wait for everything queued on the stream before it)
            if isinstance(lens, torch.Tensor) and lens.is_cuda:
                lens_src = lens.to(torch.int32)              # already uploaded by the caller (its own pinned staging ring)
            else:
                lens_src = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32)).pin_memory()
            if pool is not None and not self.can_pool(l):
                if consume is None:
                    raise ValueError("FusedBertLayers: pool needs l % 16 == 0 (pass consume= for other shapes)")
                pool = None
            key = (b, l, token_type_ids is not None, pool)
            if self.graphs > 0:
                with self._glock:
                    ent = self._graphs.get(key)
                    if ent is None:
                        if len(self._seen) > 4096:               # a corpus of ragged shapes: forget the counts, keep the graphs
                            self._seen.clear()
                        seen = self._seen[key] = self._seen.get(key, 0) + 1
                        if seen >= 2 and self._room_for_a_graph():
                            with _GATE.exclusive():
                                ent = self._graphs[key] = self._capture(b, l, token_type_ids is not None, pool)
                    if ent is not None:
                        with _GATE.shared():
                            cur = torch.cuda.current_stream(input_ids.device)
                            if ent.get("stream") is not None and ent["stream"] != cur:
                                cur.wait_stream(ent["stream"])          # a caller on another stream: order behind the last reader of the buffers
                            ent["stream"] = cur
                            self._touch(ent)
                            ent["ids"].copy_(input_ids, non_blocking=True)
                            ent["lens"].copy_(lens_src, non_blocking=True)
                            if token_type_ids is not None:
                                ent["tt"].copy_(token_type_ids, non_blocking=True)
                            ent["graph"].replay()
                            if pool is not None:
                                return ent["hidden"].clone()         # [b, hidden] fp32: the graph's static output, copied out under the lock
                            return consume(ent["hidden"]) if consume is not None else ent["hidden"].clone()
            with _GATE.shared():
                hidden = self._stack(input_ids, lens_src.to(input_ids.device, non_blocking=True), token_type_ids, pool)
                return hidden if pool is not None else (consume(hidden) if consume is not None else hidden)

    def forward_ragged(self, packed_host: np.ndarray, b: int, l: int, normalize: bool = True):
        """One mini-batch from the tokenizer's ragged output: packed_host = int32 [lens (b) | offsets (b + 1) | token ids back to
        back], l = the padded width (a multiple of 16, >= the longest sequence).  Returns the pooled rows [b, hidden] fp32
        (masked mean, L2-normalised unless normalize = False).  The host ships ONE pinned array per mini-batch — no padded id /
        mask / token-type tensors are built on either side of the link; a shape seen before replays its captured hipGraph."""
        import torch
        if l % 16 or not self.can_pool(l):
            raise ValueError("forward_ragged needs a padded width that is a multiple of 16")
        n = int(packed_host.shape[0])
        with torch.no_grad():
            src = torch.from_numpy(packed_host).pin_memory()        # pinned: a pageable source makes the host wait for the stream
            key = (b, l, "ragged", bool(normalize))
            if self.graphs > 0:
                with self._glock:
                    ent = self._graphs.get(key)
                    if ent is None:
                        if len(self._seen) > 4096:
                            self._seen.clear()
                        seen = self._seen[key] = self._seen.get(key, 0) + 1
                        if seen >= 2 and self._room_for_a_graph():
                            with _GATE.exclusive():
                                ent = self._graphs[key] = self._capture_ragged(b, l, bool(normalize))
                    if ent is not None:
                        with _GATE.shared():
                            cur = torch.cuda.current_stream(self.device)
                            if ent.get("stream") is not None and ent["stream"] != cur:
                                cur.wait_stream(ent["stream"])
                            ent["stream"] = cur
                            self._touch(ent)
                            ent["packed"][:n].copy_(src, non_blocking=True)
                            ent["graph"].replay()
                            return ent["hidden"].clone()
            with _GATE.shared():
                return self._stack_ragged(src.to(self.device, non_blocking=True), b, l, bool(normalize))

    def _capture_ragged(self, b: int, l: int, normalize: bool):
        import torch
        dev = self.device
        packed = torch.zeros((2 * b + 1 + b * l,), dtype=torch.int32, device=dev)
        packed[:b] = 1                                              # one token per sequence: a valid batch for the warm-up pass
        packed[b:2 * b + 1] = torch.arange(b + 1, dtype=torch.int32, device=dev)
        ent = {"packed": packed}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._stack_ragged(packed, b, l, normalize)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._capture_stream(), capture_error_mode="thread_local"):
            ent["hidden"] = self._stack_ragged(packed, b, l, normalize)
        ent["graph"] = graph
        return ent

    def _capture_stream(self):
        """The side stream this stack captures on (one per stack: torch.cuda.graph's default is a process-wide one, shared by every
        capturing thread)."""
        import torch
        st = getattr(self, "_cap_stream", None)
        if st is None:
            st = self._cap_stream = torch.cuda.Stream(self.device)
        return st

    def _room_for_a_graph(self) -> bool:
        """Called under the lock before a capture: with the table full, the graph that was replayed longest ago goes (its static
        buffers and private pool return to PyTorch's allocator once nothing references them) — the first `graphs` shapes of a
        process must not occupy the table forever while the shapes of today's corpus run eagerly."""
        if self.graphs <= 0:
            return False
        if len(self._graphs) >= self.graphs:
            victim = min(self._graphs, key=lambda k: self._graphs[k].get("used", 0))
            st = self._graphs[victim].get("stream")
            if st is not None:
                st.synchronize()              # its last replay (and the reader of its output) are done
            del self._graphs[victim]
        return True

    def _touch(self, ent) -> None:
        self._clock = getattr(self, "_clock", 0) + 1
        ent["used"] = self._clock

    def _capture(self, b: int, l: int, has_tt: bool, pool=None):
        """Static buffers + one eager pass on a side stream (allocator and GEMM-heuristic warm-up) + the captured pass."""
        import torch
        dev = self.device
        ent = {"ids": torch.zeros((b, l), dtype=torch.int64, device=dev), "lens": torch.ones((b,), dtype=torch.int32, device=dev),
               "tt": torch.zeros((b, l), dtype=torch.int64, device=dev) if has_tt else None}
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._stack(ent["ids"], ent["lens"], ent["tt"], pool)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._capture_stream(), capture_error_mode="thread_local"):
            ent["hidden"] = self._stack(ent["ids"], ent["lens"], ent["tt"], pool)
        ent["graph"] = graph
        return ent

    def release(self) -> None:
        """Drop the captured graphs (their private memory pools go back to PyTorch's allocator)."""
        with self._glock:
            self._graphs.clear()
            self._seen.clear()
