"""The configuration fields the embedding / retrieval path reads.

The reference keeps one big dataclass (src/comorag/utils/config_utils.py:17-298); only the fields
below are read on this path.  Any object with these attributes works — in
particular the reference's own `BaseConfig` instance — because every consumer here uses getattr
with the reference default.
"""
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class BaseConfig:
    # reference fields (same names, same defaults: utils/config_utils.py:128-176 of the reference)
    embedding_model_name: str = field(default="nvidia/NV-Embed-v2", metadata={"ref": "config_utils.py:128-129"})   # the reference's default (its factory
                                                  # has no class for it and returns None, embedding_model/__init__.py:10-17: set a "bge-" model)
    embedding_batch_size: int = 32                # :132
    embedding_return_as_normalized: bool = True   # :136
    embedding_max_seq_len: int = 2048             # :140
    embedding_model_dtype: str = "auto"           # :144  (never read by the reference; honoured here)
    linking_top_k: int = 5                        # :176
    synonymy_edge_topk: int = 2047                # :152
    synonymy_edge_query_batch_size: int = 1000    # :156
    synonymy_edge_key_batch_size: int = 10000     # :160
    synonymy_edge_sim_threshold: float = 0.8      # :164
    need_cluster: bool = True
    # new, opt-in (reference behaviour by default)
    index_dtype: str = "f32"                      # "f32" | "bf16" | "f16": storage dtype of the HBM index
    device: int = 0
    num_shards: int = 1                           # > 1: the HBM indexes are row-sharded over that many shards in THIS process (MultiDeviceIndex)
    devices: Optional[list] = None                # GPU of every shard (a device may repeat: logical shards); None = round the visible devices
    index_options: Optional[dict] = None          # route selectors for the HBM indexes (cmr_index_set_option names; "append_block_rows" for sharded ones)
    store_format: str = "parquet"                 # "parquet" (reference behaviour) | "sidecar" (append-only files)
    embedding_cache_enabled: bool = False         # probed with hasattr by the reference (BGEEmbedding.py:57-61)
    embedding_cache_path: Optional[str] = None
    embedding_length_bucketing: bool = True       # group a batch_encode call's prompts into mini-batches of similar token count
    embedding_tokenizer_threads: int = 2          # host threads tokenising ahead of the forward
    embedding_bucket_window: int = 4              # length bucketing sorts within windows of this many batches (the next windows are tokenised meanwhile)
    embedding_tokenizer_processes: int = 0        # 0: threads only; -1: worker PROCESSES started by the first corpus-sized batch_encode (>= 2 windows); N: N processes (the Rust tokenizer holds the GIL)
    embedding_forward_batches: int = 1            # length-bucketed path: a forward mini-batch holds up to this many reference batches' worth of tokens
    embedding_hip_graphs: int = 24                # fused encoder: mini-batch shapes kept as captured hipGraphs (0 = launch every forward eagerly)
    embedding_fused_encoder: bool = True          # 16-bit BERT encoders: HIP attention + bias/residual/LayerNorm stages, one QKV GEMM (embedding_model/fused_bert.py)
    embedding_devices: Optional[list] = None      # GPUs of the corpus-encode replicas (BGEEmbedding.py:77 `device_map="auto"`): one copy of the layer stack each, bucketing windows dealt round them; None = `device` only
    embedding_encode_replicas: int = 0            # > the device count: logical replicas going round `embedding_devices` (rehearsal on a one-GPU box); 0 = one per device
    embedding_query_cache: int = 256              # single-string batch_encode results kept (by prompt, max_length, normalisation): a question is encoded three times per tri_retrieve — the same prompt each time; 0 = off
    embedding_gelu: str = "exact"                 # fused encoder: "exact" = erf-form GELU kernel (the reference's function); "epilogue" = opt-in: FFN-up bias + GELU inside the hipBLASLt GEMM (TANH form, <= 4.8e-4 per activation away, ~7 % faster forward)


def cfg_get(cfg, name, default):
    return getattr(cfg, name, default) if cfg is not None else default
