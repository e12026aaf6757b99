"""OpenAIEmbeddingModel against a mocked client — CPU tier.  src/comorag/embedding_model/OpenAI.py:77-128: what goes over
the wire (newline / empty-string clean-up, batching by embedding_batch_size, the model name), what comes back (float64,
normalised iff embedding_return_as_normalized), and — when the reference tree is present — the reference class itself on the
same fake client, output for output."""
import types

import numpy as np
import pytest

from oracle.ref_loader import reference_available


class FakeClient:
    """client.embeddings.create(input=[...], model=...) -> .data[i].embedding = list of Python floats (what the SDK returns)."""

    def __init__(self, dim=24, fail_on=None):
        self.dim, self.calls, self.fail_on = dim, [], fail_on
        self.embeddings = types.SimpleNamespace(create=self._create)

    def _vec(self, text):
        import hashlib
        seed = int.from_bytes(hashlib.md5(text.encode()).digest()[:8], "little")
        return (np.random.default_rng(seed).standard_normal(self.dim) * 3.0).tolist()

    def _create(self, input, model):        # noqa: A002 (the SDK's keyword)
        self.calls.append((list(input), model))
        if self.fail_on is not None and len(self.calls) == self.fail_on:
            raise RuntimeError("HTTP 500")
        return types.SimpleNamespace(data=[types.SimpleNamespace(embedding=self._vec(t)) for t in input])


def _cfg(**kw):
    base = dict(embedding_model_name="text-embedding-3-small", embedding_batch_size=4, embedding_return_as_normalized=True,
                embedding_max_seq_len=2048, azure_embedding_endpoint=None, embedding_base_url="http://x", embedding_api_key="k")
    base.update(kw)
    return types.SimpleNamespace(**base)


TEXTS = ["a glass slipper", "line one\nline two", "", "the ball at midnight", "two stepsisters", "a pumpkin coach", "white doves",
         "the prince's search", "a hazel tree"]


def test_factory_request_shape_dtype_and_norm():
    from comorag_amd.embedding_model import _get_embedding_model_class
    cls = _get_embedding_model_class("text-embedding-3-small")
    fc = FakeClient()
    em = cls(global_config=_cfg(), embedding_model_name="text-embedding-3-small", client=fc)
    out = em.batch_encode(TEXTS)
    assert out.dtype == np.float64 and out.shape == (9, 24)                   # OpenAI.py:83: np.array of Python floats
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, rtol=0, atol=1e-12)
    assert [len(c[0]) for c in fc.calls] == [4, 4, 1] and all(c[1] == "text-embedding-3-small" for c in fc.calls)
    sent = [t for c in fc.calls for t in c[0]]
    assert sent[1] == "line one line two" and sent[2] == " "                  # OpenAI.py:78-79
    raw = np.array([fc._vec(t) for t in sent])
    np.testing.assert_array_equal(out, (raw.T / np.linalg.norm(raw, axis=1)).T)
    # one call when everything fits a batch; a str is one text; the batch size may come with the call
    fc.calls.clear()
    one = em.batch_encode("midnight")
    assert one.shape == (1, 24) and fc.calls == [(["midnight"], "text-embedding-3-small")]
    fc.calls.clear()
    em.batch_encode(TEXTS, batch_size=100, instruction="Given a question, retrieve", norm=False)
    assert len(fc.calls) == 1
    # not normalised when the configuration says so (the norm= keyword is ignored, as in the reference)
    em2 = cls(global_config=_cfg(embedding_return_as_normalized=False), embedding_model_name="text-embedding-3-small", client=fc)
    out2 = em2.batch_encode(TEXTS[:3], norm=True)
    np.testing.assert_array_equal(out2, np.array([fc._vec(t) for t in ["a glass slipper", "line one line two", " "]]))
    # encode(list) positional -> indexable rows (utils/memory_utils.py:176,205)
    assert em.encode(["x", "y"])[1].shape == (24,)


def test_a_failed_batch_raises_instead_of_dropping_rows():
    """The reference swallows the exception of a failed batch (OpenAI.py:109-117) and returns FEWER rows than texts — the
    store then binds every later id to the wrong vector.  Here the failure propagates."""
    from comorag_amd.embedding_model.openai_model import OpenAIEmbeddingModel
    em = OpenAIEmbeddingModel(global_config=_cfg(), client=FakeClient(fail_on=2))
    with pytest.raises(RuntimeError):
        em.batch_encode(TEXTS)


def test_store_on_top_keeps_float64_on_disk_and_casts_on_read(tmp_path):
    from comorag_amd.embedding_model.openai_model import OpenAIEmbeddingModel
    from comorag_amd.embedding_store import EmbeddingStore
    em = OpenAIEmbeddingModel(global_config=_cfg(), client=FakeClient())
    st = EmbeddingStore(em, str(tmp_path / "s"), 4, "chunk")
    st.insert_strings(TEXTS)
    import pyarrow.parquet as pq
    col = pq.read_table(st.filename).column("embedding")
    assert str(col.type.value_type) == "double"                               # the store persists what it was handed
    assert st.get_embeddings(st.get_all_ids()[:2]).dtype == np.float32


@pytest.mark.skipif(not reference_available(), reason="reference tree not present")
def test_equals_the_reference_class_on_the_same_client():
    from oracle.ref_loader import ref_modules
    from comorag_amd.embedding_model.openai_model import OpenAIEmbeddingModel
    import importlib
    ref_modules()
    Ref = importlib.import_module("src.comorag.embedding_model.OpenAI").OpenAIEmbeddingModel
    for norm in (True, False):
        cfg = _cfg(embedding_return_as_normalized=norm)
        fa, fb = FakeClient(), FakeClient()
        ref = Ref.__new__(Ref)                          # the constructor builds a real OpenAI client (stubbed module here)
        ref.global_config, ref.embedding_model_name = cfg, cfg.embedding_model_name
        ref._init_embedding_config()
        ref.client = fa
        ours = OpenAIEmbeddingModel(global_config=cfg, client=fb)
        for texts in (TEXTS, TEXTS[:3], "midnight"):
            a, b = ref.batch_encode(texts), ours.batch_encode(texts)
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert fa.calls == fb.calls
        assert dict(ref.embedding_config.encode_params) == dict(ours.embedding_config.encode_params) and ref.embedding_config.norm == ours.embedding_config.norm
