"""The encoder stack behind `self.embedding_model(**inputs)` (embedding_model/BGEEmbedding.py:119) for 16-bit BERT encoders.

transformers' BertLayer is ~20 kernels and as many Python module calls per layer: three projection GEMMs, SDPA with an additive
mask, the output GEMM, bias / residual adds, LayerNorm, the FFN GEMMs and GELU.  Here a layer is

    qkv  = x @ [Wq | Wk | Wv]^T + b          one hipBLASLt GEMM instead of three (PyTorch-ROCm, as north_star prescribes)
    ctx  = cmr_encoder_attention(qkv, lens)   HIP: masked softmax(QK^T/8)V straight off the packed projection, no head transposes
    x    = cmr_encoder_add_layernorm(ctx @ Wo^T, bo, x)       HIP: dense bias + residual + LayerNorm in one pass
    x    = cmr_encoder_add_layernorm(gelu(x @ W1^T + b1) @ W2^T, b2, x)

seven host calls per layer, so the thread that launches the forward leaves the interpreter lock to the tokenizer threads sooner.
The weights are the loaded model's own tensors (query / key / value concatenated once); embeddings stay the model's
`embeddings` module.  Results equal the transformers forward up to 16-bit rounding (the fused LayerNorm rounds once instead of
three times): tests/test_encoder_fused_gpu.py compares both with the fp32 oracle.

Used when `why_not(model)` is None: BERT architecture, absolute positions, exact GELU, 64-wide heads, bf16 / fp16 weights; any
other model keeps the transformers forward (`HipBGEEmbeddingModel.encoder_path` says which one runs).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib as L


def why_not(model) -> Optional[str]:
    """None if the fused layer stack can run this model, else the reason it cannot."""
    import torch
    cfg = getattr(model, "config", None)
    if cfg is None or getattr(cfg, "model_type", "") != "bert":
        return "not a BERT encoder"
    if not (hasattr(model, "embeddings") and hasattr(model, "encoder") and hasattr(model.encoder, "layer")):
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


class FusedBertLayers:
    def __init__(self, model):
        import torch
        reason = why_not(model)
        if reason is not None:
            raise ValueError("FusedBertLayers: " + reason)
        self.model = model
        cfg = model.config
        self.hidden, self.n_heads, self.eps = int(cfg.hidden_size), int(cfg.num_attention_heads), float(cfg.layer_norm_eps)
        self.dtype = next(model.parameters()).dtype
        self.cmr_dtype = L.CMR_BF16 if self.dtype == torch.bfloat16 else L.CMR_F16
        self.device = next(model.parameters()).device
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

    # ------------------------------------------------------------------ forward
    def __call__(self, input_ids, lens: np.ndarray, token_type_ids=None):
        """input_ids [b, l] int64 on the GPU (right-padded), lens[b] real token counts (host) → last hidden state [b, l, hidden]."""
        import torch
        import torch.nn.functional as F
        b, l = input_ids.shape
        with torch.no_grad():
            lens_dev = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32)).to(input_ids.device, non_blocking=True)
            x = self.model.embeddings(input_ids=input_ids, token_type_ids=token_type_ids).reshape(b * l, self.hidden)
            if not x.is_contiguous():
                x = x.contiguous()
            stream = torch.cuda.current_stream(input_ids.device).cuda_stream
            for (wqkv, bqkv, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2) in self.layers:
                qkv = F.linear(x, wqkv, bqkv)
                ctx = self.attention(qkv, lens_dev, b, l, stream)
                x = self.add_layernorm(F.linear(ctx, wo), bo, x, g1, be1, stream)
                h = F.gelu(F.linear(x, w1, b1))
                x = self.add_layernorm(F.linear(h, w2), b2, x, g2, be2, stream)
            return x.view(b, l, self.hidden)
