"""EmbeddingStore — drop-in for src/comorag/embedding_store.py:13-167 with an HBM-resident mirror.

Same constructor, methods, attributes and return values as the reference class (the fixtures in
tests/golden/store.json were produced by the reference and are replayed against this class), so
ComoRAG.py / timeline_utils.py / cluster_utils.py run on it unchanged.  What differs underneath:

* rows live in ONE growable fp32 matrix (amortised doubling) instead of a Python list of arrays:
  `get_embeddings` gathers rows without first copying all N rows (reference :150-157);
* the parquet file keeps the reference schema {hash_id: str, content: str, embedding: list<float>}
  and is written straight from the matrix with pyarrow (no per-row Python objects);
* `device_index()` lazily builds, and `insert_strings` then keeps appending to, a `DenseIndex`
  (MFMA-fragment-major panels in HBM) whose row ids equal `hash_id_to_idx` — the retrieval hooks
  (comorag_amd/hooks.py) search it instead of re-materialising host matrices.

* optional append-only persistence (`persist="sidecar"`, or `global_config.store_format`): the
  reference rewrites the whole parquet file and rebuilds four dicts on every insert
  (embedding_store.py:109-120 → O(N) per insert, O(N^2) for the probe loop's incremental appends).
  Sidecar mode appends to `vdb_{ns}.rows.jsonl` (hash_id, content) + `vdb_{ns}.f32` (raw fp32
  matrix; `vdb_{ns}.meta.json` holds the dim) and updates the dicts incrementally; a load cuts off whatever a crash
  left behind the last complete (id line, vector) pair; `export_parquet()` writes the reference-schema file
  on demand, and an existing `vdb_{ns}.parquet` is imported on first load.

Conscious deviations: `hash_id_to_text` / `text_to_hash_id` exist on an empty store too (the
reference leaves them undefined until the first save, embedding_store.py:106-107).
"""
from __future__ import annotations

import logging
import os
import threading
from copy import deepcopy
from typing import Dict, List, Optional

import numpy as np

from .utils.misc_utils import compute_mdhash_id

logger = logging.getLogger(__name__)


class _RowList:
    """List-like view of the store's matrix rows (the reference exposes `embeddings` as a list of
    ndarrays; callers index, iterate and len() it)."""

    def __init__(self, store):
        self._s = store

    def __len__(self):
        return self._s._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._s._mat[j] for j in range(*i.indices(self._s._n))]
        if i < 0:
            i += self._s._n
        if not 0 <= i < self._s._n:
            raise IndexError(i)
        return self._s._mat[i]

    def __iter__(self):
        for i in range(self._s._n):
            yield self._s._mat[i]


class EmbeddingStore:
    def __init__(self, embedding_model, db_filename, batch_size, namespace, persist: Optional[str] = None):
        self.embedding_model = embedding_model
        self.batch_size = batch_size
        self.namespace = namespace
        cfg = getattr(embedding_model, "global_config", None)
        self.persist = persist or getattr(cfg, "store_format", None) or "parquet"
        if self.persist not in ("parquet", "sidecar"):
            raise ValueError(f"persist must be 'parquet' or 'sidecar', got {self.persist!r}")
        if not os.path.exists(db_filename):
            logger.info(f"Creating working directory: {db_filename}")
            os.makedirs(db_filename, exist_ok=True)
        self.filename = os.path.join(db_filename, f"vdb_{self.namespace}.parquet")
        self._rows_file = os.path.join(db_filename, f"vdb_{self.namespace}.rows.jsonl")
        self._mat_file = os.path.join(db_filename, f"vdb_{self.namespace}.f32")
        self._meta_file = os.path.join(db_filename, f"vdb_{self.namespace}.meta.json")
        self._lock = threading.RLock()
        self._mat = np.empty((0, 0), dtype=np.float32)
        self._n = 0
        self._src_dtype = np.float32          # dtype the encoder handed over (persisted as is)
        self._index = None
        self._index_kw: dict = {}
        self._load_data()

    # ------------------------------------------------------------------ storage
    @property
    def embeddings(self):
        return _RowList(self)

    def _reserve(self, extra: int, dim: int) -> None:
        if self._mat.shape[1] != dim:
            if self._n:
                raise ValueError(f"embedding dim changed: {self._mat.shape[1]} -> {dim}")
            self._mat = np.empty((max(extra, 16), dim), dtype=np.float32)
        if self._n + extra > self._mat.shape[0]:
            cap = max(self._n + extra, 2 * self._mat.shape[0], 16)
            new = np.empty((cap, dim), dtype=np.float32)
            new[:self._n] = self._mat[:self._n]
            self._mat = new

    def _rebuild_maps(self) -> None:
        self.hash_id_to_idx = {h: idx for idx, h in enumerate(self.hash_ids)}
        self.hash_id_to_row = {h: {"hash_id": h, "content": t} for h, t in zip(self.hash_ids, self.texts)}
        self.hash_id_to_text = {h: t for h, t in zip(self.hash_ids, self.texts)}
        self.text_to_hash_id = {t: h for h, t in zip(self.hash_ids, self.texts)}

    def _load_sidecar(self) -> bool:
        """rows.jsonl (one [hash_id, content] per line) + .f32 (raw fp32 matrix) + .meta.json ({"dim": d}).
        Crash safety: an append writes the matrix rows first, then the id lines, so after a crash the files can hold
        (a) whole vectors without an id line, (b) a torn last vector, (c) a torn last id line.  The id lines that are
        complete AND have their vector are the store; everything behind them is cut off both files."""
        import json
        if not os.path.exists(self._rows_file):
            # The very first append crashed between the vectors and the id lines: whole orphan vectors and no id file.
            # They are zero valid rows — left in place, the next append would write behind them and the following load
            # would bind every id to an orphan vector.
            if os.path.exists(self._mat_file):
                os.remove(self._mat_file)
            return False
        if not os.path.exists(self._mat_file):
            return False
        rows, good_bytes = [], 0
        with open(self._rows_file, "rb") as f:
            for line in f:
                if not line.endswith(b"\n"):
                    break                      # torn last line
                try:
                    h, t = json.loads(line.decode("utf-8"))
                except Exception:
                    break
                rows.append((h, t)); good_bytes += len(line)
        n_float = os.path.getsize(self._mat_file) // 4
        dim = 0
        if os.path.exists(self._meta_file):
            dim = int(json.load(open(self._meta_file)).get("dim", 0))
        elif rows and n_float % len(rows) == 0:          # files of the first sidecar version carried no meta file
            dim = n_float // len(rows)
        if rows and dim <= 0:
            raise IOError(f"{self._mat_file}: cannot tell the embedding dim ({n_float} floats, {len(rows)} rows, no {self._meta_file})")
        n = min(len(rows), n_float // dim) if dim else 0
        if n < len(rows):                      # id lines whose vectors never made it (cannot happen with this writer's order)
            rows = rows[:n]
            good_bytes = sum(len((json.dumps([h, t], ensure_ascii=False) + "\n").encode("utf-8")) for h, t in rows)
        if os.path.getsize(self._rows_file) != good_bytes:
            with open(self._rows_file, "r+b") as f:
                f.truncate(good_bytes)
        if os.path.getsize(self._mat_file) != n * dim * 4:
            with open(self._mat_file, "r+b") as f:
                f.truncate(n * dim * 4)        # orphan tail vectors are dropped, never re-interpreted
        self.hash_ids = [h for h, _ in rows]
        self.texts = [t for _, t in rows]
        if n:
            self._reserve(n, dim)
            self._mat[:n] = np.fromfile(self._mat_file, dtype=np.float32, count=n * dim).reshape(n, dim)
            self._n = n
        return True

    def _append_sidecar(self, hash_ids, texts, rows: np.ndarray) -> None:
        import json
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        if not os.path.exists(self._meta_file):
            tmp = self._meta_file + ".tmp"
            with open(tmp, "w") as f:
                json.dump({"dim": int(rows.shape[1]), "dtype": "float32", "format": 2}, f)
                f.flush(); os.fsync(f.fileno())
            os.replace(tmp, self._meta_file)
        with open(self._mat_file, "ab") as f:          # matrix first: a crash leaves vectors without id lines (cut off on
            rows.tofile(f)                             # load), never an id line without its vector
            f.flush(); os.fsync(f.fileno())
        with open(self._rows_file, "a", encoding="utf-8") as f:
            for h, t in zip(hash_ids, texts):
                f.write(json.dumps([h, t], ensure_ascii=False) + "\n")
            f.flush(); os.fsync(f.fileno())

    def export_parquet(self, path: Optional[str] = None) -> str:
        """Write the reference-schema parquet (`hash_id`, `content`, `embedding: list<float>`)."""
        keep = self.filename
        if path is not None:
            self.filename = path
        try:
            self._write_parquet()
        finally:
            out, self.filename = self.filename, keep
        return out

    def _load_data(self):
        """embedding_store.py:92-107.  Reads files written by the reference or by this class."""
        self.hash_ids, self.texts = [], []
        if self.persist == "sidecar" and self._load_sidecar():
            self._rebuild_maps()
            logger.info(f"Loaded {len(self.hash_ids)} records from {self._rows_file}")
            return
        if os.path.exists(self.filename):
            import pyarrow.parquet as pq
            t = pq.read_table(self.filename, columns=["hash_id", "content", "embedding"])
            self.hash_ids = t.column("hash_id").to_pylist()
            self.texts = t.column("content").to_pylist()
            col = t.column("embedding").combine_chunks()
            n = len(self.hash_ids)
            if n:
                flat = col.flatten().to_numpy(zero_copy_only=False)
                self._src_dtype = flat.dtype
                dim = len(flat) // n
                assert dim * n == len(flat), "ragged embedding column"
                self._reserve(n, dim)
                self._mat[:n] = flat.reshape(n, dim).astype(np.float32, copy=False)
                self._n = n
            assert len(self.hash_ids) == len(self.texts) == self._n
            logger.info(f"Loaded {len(self.hash_ids)} records from {self.filename}")
            if self.persist == "sidecar" and self._n:      # import an existing reference file once
                for stale in (self._mat_file, self._rows_file, self._meta_file):     # e.g. vectors of a crashed first append
                    if os.path.exists(stale):
                        os.remove(stale)
                self._append_sidecar(self.hash_ids, self.texts, self._mat[:self._n])
        self._rebuild_maps()

    def _save_data(self):
        """embedding_store.py:109-120 — same file name and schema, whole-file rewrite."""
        self._write_parquet()
        self._rebuild_maps()
        logger.info(f"Saved {len(self.hash_ids)} records to {self.filename}")

    def _write_parquet(self):
        import pyarrow as pa
        import pyarrow.parquet as pq
        n, dim = self._n, (self._mat.shape[1] if self._n else 0)
        values = pa.array(np.ascontiguousarray(self._mat[:n]).reshape(-1).astype(self._src_dtype, copy=False))
        offsets = pa.array(np.arange(0, (n + 1) * dim, dim, dtype=np.int32) if dim else np.zeros(n + 1, np.int32))
        emb = pa.ListArray.from_arrays(offsets, values)
        table = pa.table({"hash_id": pa.array(self.hash_ids, type=pa.string()),
                          "content": pa.array(self.texts, type=pa.string()), "embedding": emb})
        tmp = self.filename + ".tmp"
        pq.write_table(table, tmp)
        os.replace(tmp, self.filename)

    def _upsert(self, hash_ids, texts, embeddings, dev_rows=None):
        emb = np.asarray(embeddings)
        if emb.ndim == 1:
            emb = emb[None, :]
        if self._n == 0:
            self._src_dtype = emb.dtype if emb.dtype in (np.float32, np.float64) else np.float32
        self._reserve(len(hash_ids), emb.shape[1])
        self._mat[self._n:self._n + len(hash_ids)] = emb
        new_rows = self._mat[self._n:self._n + len(hash_ids)]
        self._n += len(hash_ids)
        self.hash_ids.extend(hash_ids)
        self.texts.extend(texts)
        if self._index is not None:
            if dev_rows is not None and hasattr(self._index, "append_dev"):
                self._index.append_dev(dev_rows)
            else:
                self._index.append(new_rows)
        logger.info("Saving new records.")
        if self.persist == "sidecar":
            self._append_sidecar(hash_ids, texts, new_rows)
            base = self._n - len(hash_ids)
            for i, (h, t) in enumerate(zip(hash_ids, texts)):      # incremental dict update: O(new rows)
                self.hash_id_to_idx[h] = base + i
                self.hash_id_to_row[h] = {"hash_id": h, "content": t}
                self.hash_id_to_text[h] = t
                self.text_to_hash_id[t] = h
        else:
            self._save_data()

    # ------------------------------------------------------------------ reference API
    def _plan(self, texts: List[str]):
        """Ids and texts that are new to the store, in first-occurrence order.  A dict keyed by md5 id collapses repeated
        texts exactly as the reference's `nodes_dict` does: the first occurrence fixes the position, a later duplicate
        only rewrites the (identical) content (embedding_store.py:47-50 / :66-69)."""
        by_id: Dict[str, str] = {}
        for text in texts:
            by_id[compute_mdhash_id(text, prefix=self.namespace + "-")] = text
        known = self.hash_id_to_row
        fresh = [(h, t) for h, t in by_id.items() if h not in known]
        return len(by_id), [h for h, _ in fresh], [t for _, t in fresh]

    def get_missing_string_hash_ids(self, texts: List[str]):
        """embedding_store.py:44-61: {hash_id: {"hash_id", "content"}} of the texts not stored yet ({} for no input)."""
        _, ids, new_texts = self._plan(texts)
        return {h: {"hash_id": h, "content": t} for h, t in zip(ids, new_texts)}

    def insert_strings(self, texts: List[str]):
        """embedding_store.py:63-90: skip known ids, ONE batch_encode call for the new texts, upsert.  Returns None for
        empty input or after inserting, {} when everything was known — as the reference does."""
        with self._lock:
            n_seen, ids, new_texts = self._plan(texts)
            if n_seen == 0:
                return
            logger.info(f"Inserting {len(ids)} new records, {n_seen - len(ids)} records already exist.")
            if not ids:
                return {}
            enc_dev = getattr(self.embedding_model, "batch_encode_dev", None)
            if self._index is not None and enc_dev is not None:
                # the HBM mirror exists: the encoder's device tensor goes into it as it is (cmr_index_append_dev); the host
                # matrix gets its copy from the same tensor — no D2H -> numpy -> H2D round trip of the new rows
                dev_rows = enc_dev(new_texts)
                if hasattr(dev_rows, "is_cuda") and dev_rows.is_cuda:
                    self._upsert(ids, new_texts, dev_rows.cpu().numpy(), dev_rows=dev_rows)
                else:
                    self._upsert(ids, new_texts, np.asarray(dev_rows))
            else:
                self._upsert(ids, new_texts, self.embedding_model.batch_encode(new_texts))

    def get_row(self, hash_id):
        return self.hash_id_to_row[hash_id]

    def get_rows(self, hash_ids, dtype=np.float32):
        if not hash_ids:
            return {}
        return {id: self.hash_id_to_row[id] for id in hash_ids}

    def get_all_ids(self):
        return deepcopy(self.hash_ids)

    def get_text_for_all_rows(self):
        return deepcopy(self.hash_id_to_row)

    def get_embedding(self, hash_id, dtype=np.float32) -> np.ndarray:
        return self._mat[self.hash_id_to_idx[hash_id]].astype(dtype)

    def get_embeddings(self, hash_ids, dtype=np.float32):
        if not hash_ids:
            return []
        indices = np.array([self.hash_id_to_idx[h] for h in hash_ids], dtype=np.intp)
        return self._mat[:self._n][indices].astype(dtype, copy=False)

    def get_hash_id_to_order(self) -> Dict[str, int]:
        return {h: idx for idx, h in enumerate(self.hash_ids)}

    # ------------------------------------------------------------------ device mirror
    def device_index(self, dtype: Optional[str] = None, device: int = 0, keep_f32: bool = False, num_shards: Optional[int] = None,
                     devices=None):
        """The store's rows as a `DenseIndex` (built on first use, appended to afterwards).
        Row id == `hash_id_to_idx[hash_id]`.  dtype default: `global_config.index_dtype` of the
        embedding model when present, else "f32" (the reference's arithmetic).  `num_shards` / `devices` (default:
        `global_config.num_shards` / `.devices`): more than one shard gives a `MultiDeviceIndex` — the same rows, row-sharded
        over the node's GPUs inside this process, same results."""
        with self._lock:
            cfg = getattr(self.embedding_model, "global_config", None)
            if dtype is None:
                dtype = getattr(cfg, "index_dtype", None) or "f32"
            if num_shards is None:
                num_shards = getattr(cfg, "num_shards", None)
            if devices is None:
                devices = getattr(cfg, "devices", None)
            want = dict(dtype=dtype, device=device, keep_f32=keep_f32, num_shards=num_shards or 1, devices=tuple(devices) if devices else None)
            if self._index is not None and self._index_kw != want:
                self._index.close()
                self._index = None
            if self._index is None:
                if self._n == 0:
                    raise ValueError("device_index() on an empty store")
                from .multi_index import make_index
                self._index = make_index(self._mat.shape[1], dtype, device=device, capacity_hint=self._n, keep_f32=keep_f32,
                                         num_shards=num_shards, devices=devices, options=getattr(cfg, "index_options", None))
                self._index.append(self._mat[:self._n])
                self._index_kw = want
            return self._index
