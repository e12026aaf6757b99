"""Host logic of the multi-device index — CPU tier (no GPU: the append routing is pure host arithmetic exported on its own,
cmr_mindex_plan_append; the drop-in glue is driven with a numpy stand-in that uses THAT routing).

    comorag_amd/csrc/multi.hip: plan_append     where appended rows go (blocks round the shards, bulk = contiguous blocks)
    comorag_amd/multi_index.py                  resolve_devices / make_index / config plumbing
    comorag_amd/hooks.py                        global_config.num_shards on a REAL reference ComoRAG instance
"""
import types

import numpy as np
import pytest

from oracle.ref_loader import reference_available


def _route_py(sizes, cur, fill, m, block_rows):
    """comorag_amd/sharded.py:ShardedIndex._route, restated on plain values: the one-process-per-GPU layout's routing."""
    sizes = list(sizes)
    out = []
    while m > 0:
        if cur is None or fill >= block_rows:
            cur, fill = min(range(len(sizes)), key=lambda r: (sizes[r], r)), 0
        n = min(m, block_rows - fill)
        out.append((cur, n))
        sizes[cur] += n
        fill += n
        m -= n
    return out, cur, fill


def test_plan_append_small_appends_follow_the_block_routing_of_the_spmd_layout():
    """For appends of up to S * block_rows rows the single-process index routes exactly like ShardedIndex._route (so the
    two layouts of one corpus are interchangeable): rows fill the open block, a new block opens on the shortest shard."""
    from comorag_amd.multi_index import plan_append
    rng = np.random.default_rng(0)
    for S in (1, 2, 3, 8):
        for block in (8, 64, 8192):
            sizes = [0] * S
            cur, room = -1, 0
            pcur, pfill = None, 0
            for _ in range(60):
                m = min(int(rng.choice([1, 3, 25, 25, block, block + 1, S * block])), S * block)
                got, cur, room = plan_append(sizes, m, block, cur, room)
                want, pcur, pfill = _route_py(sizes, pcur, pfill, m, block)
                merged = []
                for s, n in want:                       # the library merges consecutive chunks of one shard
                    if merged and merged[-1][0] == s:
                        merged[-1] = (s, merged[-1][1] + n)
                    else:
                        merged.append((s, n))
                assert got == merged, (S, block, m, got, merged)
                assert sum(n for _, n in got) == m
                for s, n in got:
                    sizes[s] += n
                assert cur == pcur and room == block - pfill
            assert max(sizes) - min(sizes) <= S * block          # shards stay within a round of blocks of each other


def test_plan_append_bulk_is_contiguous_equal_blocks():
    from comorag_amd.multi_index import plan_append
    got, cur, room = plan_append([0] * 8, 10_000_000)
    assert got == [(s, 1_250_000) for s in range(8)] and room == 0
    got, _, _ = plan_append([0] * 8, 10_000_001)
    assert [n for _, n in got] == [1_250_001] * 7 + [1_249_994] and [s for s, _ in got] == list(range(8))
    # a corpus of a few thousand rows stays on ONE shard with the default block (searched without multi-shard overhead)
    got, cur, room = plan_append([0] * 8, 5_000, 65_536)
    assert got == [(0, 5_000)] and (cur, room) == (0, 60_536)
    # bulk rows on top of an unbalanced index go to the short shards first
    got, _, _ = plan_append([500_000, 0, 0, 100_000], 1_200_000, 8192, 0, 0)
    assert got[0] == (1, 300_000) and got[1] == (2, 300_000) and got[2] == (3, 300_000) and got[3] == (1, 300_000)


def test_resolve_devices_and_config_fields():
    from comorag_amd.multi_index import resolve_devices, shard_config
    from comorag_amd.utils.config_utils import BaseConfig
    assert resolve_devices(devices=[0, 0, 1]) == [0, 0, 1]
    assert resolve_devices(num_shards=4, devices=[2, 3]) == [2, 3, 2, 3]
    assert len(resolve_devices(num_shards=8)) == 8 and len(resolve_devices()) == 1
    import pytest
    for bad in (2, 3, 6):                       # would drop or unbalance GPUs of a four-device list: refused, not truncated
        with pytest.raises(ValueError):
            resolve_devices(num_shards=bad, devices=[0, 1, 2, 3])
    cfg = BaseConfig()
    assert cfg.num_shards == 1 and cfg.devices is None and cfg.index_options is None          # reference behaviour by default
    assert cfg.embedding_model_name == "nvidia/NV-Embed-v2"                                  # utils/config_utils.py:128-129 of the reference
    assert shard_config(BaseConfig(num_shards=8)) == {"num_shards": 8, "devices": None, "options": None}
    assert shard_config(None) == {"num_shards": None, "devices": None, "options": None}


class ShardedNumpyIndex:
    """numpy stand-in with MultiDeviceIndex's surface: conftest.NumpyIndex shards, rows routed by the LIBRARY's
    cmr_mindex_plan_append, global ids dense in append order, host merge.  TEST INFRASTRUCTURE (CPU tier only)."""

    def __init__(self, dim, shards, numpy_index_cls, block_rows):
        self.dim, self.n_shards, self.block_rows = dim, shards, block_rows
        self.sh = [numpy_index_cls(dim) for _ in range(shards)]
        self.gids = [[] for _ in range(shards)]
        self.total, self.cur, self.room = 0, -1, 0

    def append(self, rows):
        from comorag_amd.multi_index import plan_append
        rows = np.asarray(rows, np.float32).reshape(-1, self.dim)
        plan, self.cur, self.room = plan_append([len(s) for s in self.sh], len(rows), self.block_rows, self.cur, self.room)
        at = 0
        for s, n in plan:
            self.sh[s].append(rows[at:at + n])
            self.gids[s].extend(range(self.total, self.total + n))
            self.total += n
            at += n

    def __len__(self):
        return self.total

    def close(self):
        pass

    def shard_rows(self):
        return [len(s) for s in self.sh]

    def scores(self, q):
        q = np.asarray(q, np.float32).reshape(-1, self.dim)
        out = np.empty((len(q), self.total), np.float32)
        for s, g in zip(self.sh, self.gids):
            if g:
                out[:, g] = s.scores(q)
        return out

    def search(self, q, k, with_minmax=True):
        s = self.scores(q)
        k = min(k, s.shape[1])
        ids = np.stack([np.lexsort((np.arange(s.shape[1]), -r))[:k] for r in s])
        return ids.astype(np.int64), np.take_along_axis(s, ids, axis=1), (s.min(1) if with_minmax else None), (s.max(1) if with_minmax else None)

    def sorted_scores(self, q):
        s = self.scores(q)
        ids = np.stack([np.lexsort((np.arange(s.shape[1]), -r)) for r in s])
        return ids.astype(np.int64), np.take_along_axis(s, ids, axis=1), s.min(1), s.max(1)


@pytest.mark.skipif(not reference_available(), reason="reference tree not present")
@pytest.mark.parametrize("shards", [2, 4, 8])
def test_num_shards_of_the_config_reaches_the_indexes_of_a_real_comorag_instance(tmp_path, fake_embedder, numpy_index_cls, monkeypatch, shards):
    """hooks.install on a REAL reference ComoRAG instance whose global_config carries num_shards / devices / index_options:
    the three matrices land on sharded indexes (here: the numpy stand-in behind comorag_amd.multi_index.make_index, rows
    routed by the library's own rule) and every call the reference's tri_retrieve makes returns what the reference's own
    methods return on an identical instance (ComoRAG.py:937-967)."""
    from oracle.ref_loader import ref_modules
    from tests.test_binding_reference import _bare_rag, _stores
    m = ref_modules()
    ComoRAG = m["ComoRAG"].ComoRAG
    from comorag_amd import hooks, multi_index
    from comorag_amd.embedding_store import EmbeddingStore
    seen = []

    def fake_make_index(dim, dtype="f32", device=0, capacity_hint=0, keep_f32=False, options=None, num_shards=None, devices=None):
        devs = multi_index.resolve_devices(num_shards, devices, device)
        seen.append((len(devs), tuple(devs), dict(options or {})))
        return ShardedNumpyIndex(dim, len(devs), numpy_index_cls, (options or {}).get("append_block_rows", 65536))

    monkeypatch.setattr(multi_index, "make_index", fake_make_index)
    ours = _bare_rag(ComoRAG, _stores(tmp_path, fake_embedder, EmbeddingStore), fake_embedder)
    ref = _bare_rag(ComoRAG, _stores(tmp_path, fake_embedder, m["embedding_store"].EmbeddingStore), fake_embedder)
    ours.global_config = types.SimpleNamespace(need_cluster=True, index_dtype="f32", num_shards=shards, devices=[0] * shards,
                                               index_options={"append_block_rows": 2})
    hooks.install(ours, patch_module_functions=False)
    ours.prepare_retrieval_objects()
    ref.prepare_retrieval_objects()
    assert len(seen) == 3 and all(s[0] == shards and s[2] == {"append_block_rows": 2} for s in seen)
    # (8 chunks, 6 facts, 3 summaries in blocks of 2 rows: the rows really are spread over the shards)
    assert sum(1 for r in ours._hip["passage"].shard_rows() if r) == min(shards, 4)
    for q in ["who lost a slipper?", "what became a coach?", "who helped cinderella?"]:
        a_ids, a_sc = ours.dense_passage_retrieval(q)
        b_ids, b_sc = ref.dense_passage_retrieval(q)
        assert a_ids.tolist() == b_ids.tolist()
        np.testing.assert_allclose(a_sc, b_sc, atol=2e-6)
        np.testing.assert_allclose(ours.get_fact_scores(q), ref.get_fact_scores(q), atol=2e-6)
        c_ids, c_sc = ours.dense_passage_retrieval(q, need_cluster=True)
        d_ids, d_sc = ref.dense_passage_retrieval(q, need_cluster=True)
        assert c_ids.tolist() == d_ids.tolist()
        np.testing.assert_allclose(c_sc, d_sc, atol=2e-6)


def test_a_new_question_is_encoded_once_when_the_model_ignores_its_instruction(fake_embedder, numpy_index_cls):
    """hooks.get_query_embeddings: the reference encodes a new question once per instruction (ComoRAG.py:921-935), and its BGE model overwrites the
    instruction it is handed (BGEEmbedding.py:150-155) — two identical forwards.  An embedding model that declares the quirk
    (`instruction_is_ignored`, HipBGEEmbeddingModel) is called ONCE and both caches hold that row; any other model is called once per kind as
    before; a question already cached for one kind goes through the per-kind path."""
    import sys
    from comorag_amd import hooks
    rng = np.random.default_rng(0)
    X, F = rng.standard_normal((9, 32)).astype(np.float32), rng.standard_normal((5, 32)).astype(np.float32)

    class Rag:
        def __init__(self, emb):
            self.global_config = types.SimpleNamespace(need_cluster=False, index_dtype="f32")
            self.embedding_model = emb
            self.ready_to_retrieve = False
        def prepare_retrieval_objects(self):
            self.query_to_embedding = {"triple": {}, "passage": {}}
            self.passage_embeddings, self.fact_embeddings = X, F
            self.ready_to_retrieve = True
    sys.modules[Rag.__module__].get_query_instruction = lambda k: f"<{k}> "

    def factory(mat, dtype, device):
        ix = numpy_index_cls(np.asarray(mat).shape[1], dtype, device)
        ix.append(np.asarray(mat, np.float32))
        return ix

    plain = hooks.install(Rag(fake_embedder), index_factory=factory, patch_module_functions=False)
    plain.prepare_retrieval_objects()
    plain.get_query_embeddings("who lost a slipper?")
    assert fake_embedder.calls == [["who lost a slipper?"], ["who lost a slipper?"]]          # one call per kind: the model says nothing about its instruction

    class Quirky(type(fake_embedder)):
        instruction_is_ignored = True
    emb = Quirky(32)
    rag = hooks.install(Rag(emb), index_factory=factory, patch_module_functions=False)
    rag.prepare_retrieval_objects()
    rag.get_query_embeddings(["who lost a slipper?", "what became a coach?"])
    assert emb.calls == [["who lost a slipper?"], ["what became a coach?"]]                    # once per question
    t, p = rag.query_to_embedding["triple"], rag.query_to_embedding["passage"]
    assert all(np.array_equal(t[q], p[q]) and t[q].shape == (1, 32) for q in t) and set(t) == set(p)
    np.testing.assert_array_equal(t["what became a coach?"][0], emb._vec("what became a coach?"))
    rag.get_query_embeddings("who lost a slipper?")                                            # memoised
    rag.dense_passage_retrieval("who lost a slipper?"); rag.get_fact_scores("who lost a slipper?")
    assert len(emb.calls) == 2
    rag.query_to_embedding["triple"]["half cached"] = emb._vec("half cached")[None]
    n0 = len(emb.calls)
    rag.get_query_embeddings("half cached")
    assert emb.calls[n0:] == [["half cached"]] and "half cached" in rag.query_to_embedding["passage"]
