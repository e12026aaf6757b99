"""Host logic of the EmbeddingStore drop-in, replayed against what the REFERENCE class did
(tests/golden/store.json, store_emb.npz — produced by oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from comorag_amd.embedding_store import EmbeddingStore
from comorag_amd.utils.misc_utils import compute_mdhash_id, min_max_normalize


def test_mdhash_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "mdhash.json")))
    got = [compute_mdhash_id(s, prefix=p) for s in g["strings"] for p in ("", "chunk-", "entity-")]
    assert got == g["ids"]


def test_minmax_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "minmax.npz"))
    assert np.array_equal(min_max_normalize(g["v"]), g["v_out"])
    assert np.array_equal(min_max_normalize(g["c"]), g["c_out"])
    assert min_max_normalize(np.array([], dtype=np.float32)).size == 0


def test_store_replays_reference(golden_dir, tmp_path, fake_embedder):
    g = json.load(open(os.path.join(golden_dir, "store.json")))
    ge = np.load(os.path.join(golden_dir, "store_emb.npz"))
    st = EmbeddingStore(fake_embedder, str(tmp_path), 8, "chunk")
    assert st.filename.endswith("vdb_chunk.parquet")
    r1 = st.insert_strings(g["batch1"])
    assert list(st.hash_ids) == g["ids_after_1"]
    r2 = st.insert_strings(g["batch2"])
    r3 = st.insert_strings(["alpha"])
    r4 = st.insert_strings([])
    assert [repr(r) for r in (r1, r2, r3, r4)] == g["ret"]
    assert list(st.hash_ids) == g["ids_after_2"] and list(st.texts) == g["texts"]
    assert fake_embedder.calls == g["encode_calls"]                    # same encode batches, same order
    assert st.get_missing_string_hash_ids(["alpha", "zeta", "zeta"]) == g["missing"]
    assert st.get_hash_id_to_order() == g["hash_id_to_idx"]
    np.testing.assert_array_equal(st.get_embeddings(list(st.hash_ids)), ge["all"])
    np.testing.assert_array_equal(st.get_embedding(st.hash_ids[2]), ge["one"])
    assert st.get_embeddings([]) == [] and st.get_rows([]) == {}
    assert st.get_row(st.hash_ids[1]) == {"hash_id": st.hash_ids[1], "content": "beta"}
    assert st.text_to_hash_id["gamma"] == st.hash_ids[2] and st.hash_id_to_text[st.hash_ids[0]] == "alpha"
    assert len(st.embeddings) == 5 and st.embeddings[0].dtype == np.float32
    ids = st.get_all_ids(); ids.append("x"); assert len(st.hash_ids) == 5          # deepcopy semantics
    # reload from parquet
    st2 = EmbeddingStore(fake_embedder, str(tmp_path), 8, "chunk")
    assert list(st2.hash_ids) == g["reload_ids"] and list(st2.texts) == g["reload_texts"]
    np.testing.assert_array_equal(st2.get_embeddings(list(st2.hash_ids)), ge["all"])
    assert type(st2.embeddings[0]).__name__ == g["reload_emb_type"] and str(st2.embeddings[0].dtype) == g["reload_emb_dtype"]


def test_parquet_schema_is_reference_compatible(tmp_path, fake_embedder):
    """The file must load with the reference's reader (pd.read_parquet → three .tolist() columns,
    embedding_store.py:94-95) and the reference's own files must load here."""
    import pandas as pd
    st = EmbeddingStore(fake_embedder, str(tmp_path), 8, "ns")
    st.insert_strings(["a", "b", "c"])
    df = pd.read_parquet(st.filename)
    assert list(df.columns) == ["hash_id", "content", "embedding"]
    embs = df["embedding"].values.tolist()
    assert isinstance(embs[0], np.ndarray) and embs[0].dtype == np.float32 and len(embs[0]) == 32
    # a file written the reference's way (pandas, list of ndarrays)
    ref = pd.DataFrame({"hash_id": ["r-1", "r-2"], "content": ["x", "y"],
                        "embedding": [np.arange(4, dtype=np.float32), np.ones(4, dtype=np.float32)]})
    os.makedirs(tmp_path / "ref", exist_ok=True)
    ref.to_parquet(tmp_path / "ref" / "vdb_r.parquet", index=False)
    st3 = EmbeddingStore(fake_embedder, str(tmp_path / "ref"), 8, "r")
    assert st3.hash_ids == ["r-1", "r-2"] and np.array_equal(st3.get_embedding("r-1"), np.arange(4, dtype=np.float32))


def test_empty_store_and_float64_encoder(tmp_path):
    class F64:
        def batch_encode(self, texts, **kw):
            return np.array([[float(len(t)), 1.0] for t in texts])          # float64 like OpenAI.py:83
    st = EmbeddingStore(F64(), str(tmp_path), 8, "e")
    assert st.hash_ids == [] and st.hash_id_to_idx == {} and st.hash_id_to_row == {} and st.get_all_ids() == []
    assert st.hash_id_to_text == {} and st.text_to_hash_id == {}               # defined (conscious fix)
    st.insert_strings(["ab", "abc"])
    assert st.get_embeddings(st.get_all_ids()).dtype == np.float32
    import pandas as pd
    assert pd.read_parquet(st.filename)["embedding"].values.tolist()[0].dtype == np.float64


def test_live_reference_agrees(tmp_path, fake_embedder):
    from oracle.ref_loader import reference_available, ref_modules
    if not reference_available():
        pytest.skip("reference tree not present")
    Ref = ref_modules()["embedding_store"].EmbeddingStore
    from tests.conftest import FakeEmbedder
    a, b = Ref(FakeEmbedder(16), str(tmp_path / "a"), 4, "chunk"), EmbeddingStore(FakeEmbedder(16), str(tmp_path / "b"), 4, "chunk")
    for batch in (["x", "y", "x"], ["z", "y", "w"], [], ["w"]):
        ra, rb = a.insert_strings(batch), b.insert_strings(batch)
        assert repr(ra) == repr(rb)
        assert list(a.hash_ids) == list(b.hash_ids) and list(a.texts) == list(b.texts)
        assert a.hash_id_to_idx == b.hash_id_to_idx and a.hash_id_to_row == b.hash_id_to_row
    np.testing.assert_array_equal(np.asarray(a.get_embeddings(a.get_all_ids())), b.get_embeddings(b.get_all_ids()))


def test_make_cache_embed_roundtrip(tmp_path):
    """sqlite embedding cache (embedding_model/base.py:112-187 semantics): second call is served
    from the file, order preserved, keyword-only call like the reference wrapper."""
    import torch
    from comorag_amd.embedding_model.base import make_cache_embed
    calls = []
    def enc(**kw):
        calls.append(list(kw["prompts"]))
        return torch.tensor([[float(len(p)), 1.0, 2.0] for p in kw["prompts"]])
    f = make_cache_embed(enc, str(tmp_path / "cache.db"), "cpu")
    a = f(prompts=["aa", "b", "cccc"], instruction="I", max_length=16)
    b = f(prompts=["b", "zz", "aa"], instruction="I", max_length=16)
    assert calls == [["aa", "b", "cccc"], ["zz"]]
    assert a[:, 0].tolist() == [2.0, 1.0, 4.0] and b[:, 0].tolist() == [1.0, 2.0, 2.0]
    c = f(prompts=["aa"], instruction="other", max_length=16)          # different instruction → different key
    assert calls[-1] == ["aa"] and c.shape == (1, 3)


def test_sidecar_persistence_is_append_only_and_equivalent(tmp_path, fake_embedder):
    """persist="sidecar": same observable store as parquet mode, files only ever grow, reload and
    parquet export/import round-trip."""
    from tests.conftest import FakeEmbedder
    a = EmbeddingStore(FakeEmbedder(32), str(tmp_path / "p"), 8, "chunk")
    b = EmbeddingStore(FakeEmbedder(32), str(tmp_path / "s"), 8, "chunk", persist="sidecar")
    sizes = []
    for batch in (["x", "y", "x"], ["z", "y", "w"], [], ["w"], ["q" * 50, "naïve ☕"]):
        ra, rb = a.insert_strings(batch), b.insert_strings(batch)
        assert repr(ra) == repr(rb)
        assert a.hash_ids == b.hash_ids and a.texts == b.texts and a.hash_id_to_idx == b.hash_id_to_idx
        assert a.hash_id_to_row == b.hash_id_to_row and a.text_to_hash_id == b.text_to_hash_id
        sizes.append((os.path.getsize(b._mat_file) if os.path.exists(b._mat_file) else 0))
    assert sizes == sorted(sizes) and sizes[-1] == len(b.hash_ids) * 32 * 4
    assert not os.path.exists(b.filename)                         # no parquet rewrite in sidecar mode
    b2 = EmbeddingStore(FakeEmbedder(32), str(tmp_path / "s"), 8, "chunk", persist="sidecar")
    assert b2.hash_ids == a.hash_ids and b2.texts == a.texts
    np.testing.assert_array_equal(b2.get_embeddings(b2.get_all_ids()), a.get_embeddings(a.get_all_ids()))
    out = b2.export_parquet()
    import pandas as pd
    df = pd.read_parquet(out)                                     # reference reader
    assert df["hash_id"].tolist() == a.hash_ids and df["content"].tolist() == a.texts
    # importing an existing reference-schema parquet into sidecar mode
    c = EmbeddingStore(FakeEmbedder(32), str(tmp_path / "p"), 8, "chunk", persist="sidecar")
    assert c.hash_ids == a.hash_ids and os.path.exists(c._mat_file)
    c.insert_strings(["brand new"])
    c2 = EmbeddingStore(FakeEmbedder(32), str(tmp_path / "p"), 8, "chunk", persist="sidecar")
    assert c2.hash_ids == a.hash_ids + [compute_mdhash_id("brand new", prefix="chunk-")]


def test_sidecar_load_survives_a_crash_between_the_two_appends(tmp_path):
    """An append writes vectors first, id lines second.  A crash in between leaves whole orphan vectors (or a torn one,
    or a torn id line); the next load must come up with exactly the complete (id, vector) pairs — not raise forever, and
    never re-interpret the floats with a wrong dim (2 rows of dim 8 + 2 orphan rows used to load as 2 rows of dim 16)."""
    from tests.conftest import FakeEmbedder
    d = str(tmp_path / "s")
    a = EmbeddingStore(FakeEmbedder(8), d, 8, "chunk", persist="sidecar")
    a.insert_strings(["one", "two"])
    want_ids, want = list(a.hash_ids), np.array(a.get_embeddings(a.get_all_ids()))
    orphan = np.arange(16, dtype=np.float32)
    for tail_f32, tail_rows in ((orphan.tobytes(), b""),                                  # two whole orphan vectors
                                (orphan.tobytes()[:20], b""),                             # a torn vector
                                (orphan.tobytes(), b'["chunk-deadbeef", "thr')):          # vectors + a torn id line
        with open(a._mat_file, "ab") as f:
            f.write(tail_f32)
        with open(a._rows_file, "ab") as f:
            f.write(tail_rows)
        b = EmbeddingStore(FakeEmbedder(8), d, 8, "chunk", persist="sidecar")
        assert b.hash_ids == want_ids and b._mat.shape[1] == 8
        np.testing.assert_array_equal(b.get_embeddings(b.get_all_ids()), want)
        assert os.path.getsize(b._mat_file) == 2 * 8 * 4                  # the tail is gone: later appends line up again
    b.insert_strings(["three"])
    c = EmbeddingStore(FakeEmbedder(8), d, 8, "chunk", persist="sidecar")
    assert c.hash_ids == want_ids + [compute_mdhash_id("three", prefix="chunk-")] and len(c.embeddings) == 3
    np.testing.assert_array_equal(c.get_embedding(c.hash_ids[2]), FakeEmbedder(8)._vec("three"))
    # the very FIRST append crashed after the vectors, before any id line: orphan vectors, no id file.  The store comes up
    # empty, and what is inserted next must be bound to ITS vectors on the following load (not to the orphans)
    f0 = str(tmp_path / "first")
    os.makedirs(f0)
    with open(os.path.join(f0, "vdb_chunk.f32"), "wb") as f:
        f.write(orphan.tobytes())
    with open(os.path.join(f0, "vdb_chunk.meta.json"), "w") as f:
        f.write('{"dim": 8, "dtype": "float32", "format": 2}')
    g0 = EmbeddingStore(FakeEmbedder(8), f0, 8, "chunk", persist="sidecar")
    assert g0.hash_ids == [] and not os.path.exists(g0._mat_file)
    g0.insert_strings(["one", "two"])
    g1 = EmbeddingStore(FakeEmbedder(8), f0, 8, "chunk", persist="sidecar")
    assert g1.hash_ids == want_ids and os.path.getsize(g1._mat_file) == 2 * 8 * 4
    np.testing.assert_array_equal(g1.get_embeddings(g1.get_all_ids()), want)
    # import path: a parquet file next to stale sidecar vectors (crashed first append, no id file): the parquet rows win
    p = str(tmp_path / "p")
    EmbeddingStore(FakeEmbedder(8), p, 8, "chunk").insert_strings(["one", "two"])
    with open(os.path.join(p, "vdb_chunk.f32"), "wb") as f:
        f.write(orphan.tobytes())
    e = EmbeddingStore(FakeEmbedder(8), p, 8, "chunk", persist="sidecar")
    assert e.hash_ids == want_ids and os.path.getsize(e._mat_file) == 2 * 8 * 4
    e2 = EmbeddingStore(FakeEmbedder(8), p, 8, "chunk", persist="sidecar")
    np.testing.assert_array_equal(e2.get_embeddings(e2.get_all_ids()), want)


def test_tokenize_batch_equals_the_reference_tokenizer_call():
    """comorag_amd.embedding_model.bge.tokenize_batch must hand the encoder exactly what
    BGEEmbedding.py:112-117 does (`tokenizer(prompts, padding=True, truncation=True, max_length=...,
    return_tensors="pt")`): same keys, int64 tensors, padding to the longest prompt, truncation."""
    import torch
    from comorag_amd.embedding_model.bge import tokenize_batch
    from tools.synthetic import synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    chunks = synthetic_chunks(words, 8)
    texts = chunks[:5] + ["", "a", chunks[5][:50], "   ", chunks[6] * 3]
    for max_length in (512, 16, 3):
        ref = tok(texts, padding=True, truncation=True, max_length=max_length, return_tensors="pt")
        got = tokenize_batch(tok, texts, max_length)
        assert set(ref.keys()) == set(got.keys())
        for key in ref:
            assert got[key].dtype == ref[key].dtype == torch.int64 and torch.equal(got[key], ref[key]), key
    one = tokenize_batch(tok, [texts[0]], 512)
    assert one["input_ids"].ndim == 2 and one["input_ids"].shape[0] == 1


def test_length_bucketing_builds_the_tokenizers_own_tensors():
    """tokenize_ragged + pad_batch (length-bucketed mini-batches) must give, for any group of prompts, exactly the
    tensors `tokenizer(group, padding=True, truncation=True, max_length=..., return_tensors="pt")` gives."""
    import torch
    from comorag_amd.embedding_model.bge import pad_batch, tokenize_ragged
    from tools.synthetic import synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    chunks = synthetic_chunks(words, 6)
    texts = [chunks[0][:40], chunks[1], "", "a", chunks[2][:300], chunks[3] * 2, "  ", chunks[4][:90]]
    for max_length in (512, 24):
        rag = tokenize_ragged(tok, texts, max_length)
        order = np.argsort([len(x) for x in rag], kind="stable")
        for g in (order[:3], order[3:6], order[6:]):
            ref = tok([texts[j] for j in g], padding=True, truncation=True, max_length=max_length, return_tensors="pt")
            got = pad_batch(tok, [rag[j] for j in g])
            assert set(got.keys()) == set(ref.keys())
            for key in ref:
                assert torch.equal(got[key], ref[key]), (key, max_length)


def test_tokenizer_worker_process_returns_what_the_in_process_tokenizer_returns():
    """embedding_tokenizer_processes > 0: the spawned, tokenizers-only worker (comorag_amd/embedding_model/_tokworker.py)
    must hand back exactly the id lists `tokenize_ragged` builds in-process — same truncation, same special tokens."""
    import multiprocessing as mp
    from comorag_amd.embedding_model import _tokworker
    from comorag_amd.embedding_model.bge import tokenize_ragged
    from tools.synthetic import synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    texts = synthetic_chunks(words, 6, tokens_per_chunk=560) + ["", "a", "   ", "short one"]
    want = tokenize_ragged(tok, texts, 512)
    assert max(len(x) for x in want) == 512 and min(len(x) for x in want) == 2
    _tokworker.init(tok.backend_tokenizer.to_str())
    as_lists = lambda arrays: [a.tolist() for a in arrays]
    assert as_lists(_tokworker.ragged(texts, 512)) == want
    assert as_lists(_tokworker.ragged(texts[:3], 64)) == tokenize_ragged(tok, texts[:3], 64)
    with mp.get_context("spawn").Pool(1, initializer=_tokworker.init, initargs=(tok.backend_tokenizer.to_str(),)) as pool:
        got = pool.apply(_tokworker.ragged, (texts, 512))
        assert as_lists(got) == want and all(a.dtype == np.int32 for a in got)


def test_private_backend_copies_equal_the_tokenizer_call_under_concurrency():
    """HipBGEEmbeddingModel tokenises through private, never-reconfigured copies of the Rust backend (one per max_length):
    same ids / padded tensors as `tokenizer(prompts, padding=..., truncation=True, max_length=...)` (BGEEmbedding.py:112-117),
    also when threads ask for different lengths at the same time (the shared backend is re-configured per call and races)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel, tokenize_batch, tokenize_ragged
    from tools.synthetic import synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    texts = synthetic_chunks(words, 12, tokens_per_chunk=560) + ["midnight", "the prince and the bird", ""]
    em = HipBGEEmbeddingModel.__new__(HipBGEEmbeddingModel)                 # host-side methods only: no GPU in this tier
    em.tokenizer, em._fast_tok, em._bt_copies, em._bt_lock, em.max_positions = tok, True, {}, threading.Lock(), 512
    want = {ml: (tokenize_ragged(tok, texts, ml), tokenize_batch(tok, texts, ml)) for ml in (512, 64, 7)}

    def one(ml):
        rag, pad = em._ragged(texts, ml), em._tokenize(texts, ml)
        assert all(a.dtype == np.int32 for a in rag)                        # arrays, made on the tokenizer's thread
        return [a.tolist() for a in rag] == want[ml][0] and set(pad) == set(want[ml][1]) and all(
            pad[k].dtype == want[ml][1][k].dtype and (pad[k].numpy() == want[ml][1][k].numpy()).all() for k in pad)
    with ThreadPoolExecutor(6) as ex:
        assert all(ex.map(one, [512, 64, 7] * 8))
    assert sorted(em._bt_copies) == [7, 64, 512]


def test_fused_encoder_gate_and_mask_lengths():
    """Host logic of the 16-bit layer stack (embedding_model/fused_bert.py): which models it accepts, and that only
    ones-then-zeros attention masks are turned into per-sequence lengths (anything else keeps the transformers forward)."""
    import torch
    from comorag_amd.embedding_model import fused_bert
    from oracle import encode_torch as enc
    m32, _ = enc.tiny_bert(hidden=128, layers=1, heads=4, inter=256, max_pos=32)      # 32-wide heads
    m64, _ = enc.tiny_bert(hidden=256, layers=1, heads=4, inter=512, max_pos=32)
    assert "16-bit" in fused_bert.why_not(m64) and "head width" in fused_bert.why_not(m32.to(torch.bfloat16))
    assert fused_bert.why_not(m64.to(torch.bfloat16)) is None and fused_bert.why_not(m64.to(torch.float16)) is None
    m64.config.hidden_act = "relu"
    assert "activation" in fused_bert.why_not(m64)
    assert fused_bert.why_not(object()) == "not a BERT / RoBERTa / XLM-R encoder"
    mx, _ = enc.tiny_xlmr(hidden=128, layers=1, heads=2, inter=128, max_pos=66)
    assert fused_bert.why_not(mx.to(torch.bfloat16)) is None and fused_bert.position_offset(mx) == 2 and fused_bert.position_offset(m64) == 0
    assert fused_bert.lens_of_mask(np.array([[1, 1, 0, 0], [1, 1, 1, 1]])).tolist() == [2, 4]
    for bad in ([[0, 1, 1]], [[1, 0, 1]], [[0, 0, 0]], [[1, 2, 0]], [1, 1, 0]):
        assert fused_bert.lens_of_mask(np.array(bad)) is None
