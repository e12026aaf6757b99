"""The HIP stages of the 16-bit BERT layer stack (comorag_amd/csrc/encoder_kernels.hip, embedding_model/fused_bert.py) against
fp32 restatements on the same rounded inputs, and the whole encoder against the transformers forward and the fp32 oracle."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {"bfloat16": 2e-2, "float16": 2.5e-3}


class _Stages:
    """FusedBertLayers' two kernel wrappers without a model around them."""
    def __init__(self, hidden, heads, dtype, eps=1e-12):
        import torch
        from comorag_amd import _lib as L
        from comorag_amd.embedding_model.fused_bert import FusedBertLayers
        self.hidden, self.n_heads, self.eps = hidden, heads, eps
        self.cmr_dtype = L.CMR_BF16 if dtype == torch.bfloat16 else L.CMR_F16
        self.attention = FusedBertLayers.attention.__get__(self)
        self.add_layernorm = FusedBertLayers.add_layernorm.__get__(self)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("shape", [(3, 100, 2), (2, 512, 12), (5, 37, 4), (2, 129, 1), (3, 300, 16), (1, 64, 3)])
def test_attention_kernel_vs_fp32_softmax(dtype, shape):
    import torch
    b, l, heads = shape
    hidden, tdt = heads * 64, getattr(torch, dtype)
    g = torch.Generator(device="cuda").manual_seed(b * 1000 + l)
    qkv = (torch.randn((b * l, 3 * hidden), generator=g, device="cuda") * 1.5).to(tdt)          # peaked, non-symmetric scores
    lens = np.random.default_rng(l).integers(1, l + 1, size=b).astype(np.int32)
    lens[0] = l
    if b > 1:
        lens[1] = 1 if l < 200 else l - 130                                                       # a one-token row / a block of padding only
    lens_dev = torch.from_numpy(lens).cuda()
    got = _Stages(hidden, heads, tdt).attention(qkv, lens_dev, b, l).float().view(b, l, heads, 64)
    torch.cuda.synchronize()
    x = qkv.float().view(b, l, 3, heads, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))                                  # [b, heads, l, 64]
    scores = q @ k.transpose(-1, -2) / 8.0
    keymask = torch.arange(l, device="cuda")[None, :] >= lens_dev[:, None]                        # [b, l] True = padding
    scores = scores.masked_fill(keymask[:, None, None, :], float("-inf"))
    want = (torch.softmax(scores, dim=-1) @ v).permute(0, 2, 1, 3)                                # [b, l, heads, 64]
    assert torch.isfinite(got).all()
    for s in range(b):
        np.testing.assert_allclose(got[s, :lens[s]].cpu().numpy(), want[s, :lens[s]].cpu().numpy(), atol=TOL[dtype], rtol=TOL[dtype])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("d,rows", [(768, 1001), (1024, 64), (256, 7), (40, 130), (2048, 33), (1536, 5)])
@pytest.mark.parametrize("parts", ["bias+residual", "residual", "plain"])
def test_add_layernorm_kernel(dtype, d, rows, parts):
    import torch
    tdt = getattr(torch, dtype)
    g = torch.Generator(device="cuda").manual_seed(d + rows)
    rnd = lambda *s: torch.randn(s, generator=g, device="cuda")
    y, res = (rnd(rows, d) * 2 + 0.3).to(tdt), rnd(rows, d).to(tdt)
    bias, gamma, beta = rnd(d).to(tdt), (1 + 0.2 * rnd(d)).to(tdt), (0.1 * rnd(d)).to(tdt)
    use_b, use_r = parts == "bias+residual", parts != "plain"
    got = _Stages(d, 1, tdt, eps=1e-12).add_layernorm(y, bias if use_b else None, res if use_r else None, gamma, beta).float()
    z = y.float() + (bias.float() if use_b else 0) + (res.float() if use_r else 0)
    want = torch.nn.functional.layer_norm(z, (d,), gamma.float(), beta.float(), 1e-12)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=TOL[dtype], rtol=TOL[dtype])


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_embedding_kernel_vs_transformers_module(dtype):
    import torch
    from comorag_amd.embedding_model.fused_bert import FusedBertLayers
    from oracle import encode_torch as enc
    model, _ = enc.tiny_bert(hidden=256, layers=1, heads=4, inter=512, max_pos=96)
    with torch.no_grad():
        for p in model.embeddings.parameters():
            p.mul_(30.0).add_(0.05)                   # init std 0.02: lift the tables clear of the 16-bit rounding floor
    model = model.to("cuda", dtype=getattr(torch, dtype)).eval()
    fz = FusedBertLayers(model)
    g = torch.Generator(device="cuda").manual_seed(7)
    ids = torch.randint(0, model.config.vocab_size, (5, 96), generator=g, device="cuda")
    tt = torch.randint(0, 2, (5, 96), generator=g, device="cuda")
    ref = model.float()                                # fp32 arithmetic on the same 16-bit-representable tables
    for types in (None, tt):
        got = fz.embed(ids, types).float().view(5, 96, 256)
        with torch.no_grad():
            want = ref.embeddings(input_ids=ids, token_type_ids=types)
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=TOL[dtype], rtol=TOL[dtype])
    wide = fz.embed(torch.full((1, 3), 10 ** 9, device="cuda", dtype=torch.int64))      # out-of-table ids are clamped, not a fault
    assert torch.isfinite(wide.float()).all()


def _peaked_tiny_bert(dtype):
    import torch
    from oracle import encode_torch as enc
    model, tok = enc.tiny_bert(hidden=256, layers=3, heads=4, inter=512, max_pos=128)
    with torch.no_grad():                              # random-init scores are ~0 (uniform attention): make the softmax matter
        for lyr in model.encoder.layer:
            lyr.attention.self.query.weight.mul_(12.0)
            lyr.attention.self.key.weight.mul_(12.0)
        model.to(dtype).float()                        # the oracle runs fp32 arithmetic on the SAME 16-bit-representable weights
    return model, tok


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_fused_layer_stack_vs_transformers_forward_and_oracle(dtype):
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = _peaked_tiny_bert(getattr(torch, dtype))
    texts = [f"the prince and the golden slipper number {i} " + "and the bird in the tree " * (i % 7) for i in range(23)]
    texts += ["she was good and pious " * 40, "midnight"]
    want = enc.batch_encode(model, tok, texts, batch_size=8, max_length=128)
    cls = _get_embedding_model_class("bge-tiny-random")
    ems = {}
    for fused in (True, False):
        cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128,
                         embedding_model_dtype=dtype, embedding_fused_encoder=fused)
        ems[fused] = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    assert ems[True].encoder_path == "hip-fused-layers" and ems[False].encoder_path == "transformers"
    got, plain = ems[True].batch_encode(texts), ems[False].batch_encode(texts)
    tol = {"bfloat16": 6e-3, "float16": 1e-3}[dtype]                     # unit-norm rows of 256: components ~0.06
    np.testing.assert_allclose(got, want, atol=tol)
    np.testing.assert_allclose(plain, want, atol=tol)
    assert np.abs(got - want).max() <= 2.0 * np.abs(plain - want).max() + 2e-4     # no worse than the transformers forward in the same dtype
    assert float(np.min((got * want).sum(1))) > 0.9995
    # a single string (one padded row of its own length) and the reference's arrival-order mini-batches
    np.testing.assert_allclose(ems[True].batch_encode("midnight"), want[-1:], atol=tol)
    cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128, embedding_model_dtype=dtype,
                     embedding_length_bucketing=False)
    em = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    np.testing.assert_allclose(em.batch_encode(texts), want, atol=tol)
    for e in (*ems.values(), em):
        e.close()


def test_models_the_fused_stack_declines_keep_the_transformers_forward():
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.embedding_model import fused_bert
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = enc.tiny_bert(hidden=128, layers=1, heads=4, inter=256, max_pos=64)     # 32-wide heads
    assert "head width" in fused_bert.why_not(copy.deepcopy(model).to(torch.bfloat16))
    cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_model_dtype="bf16")
    em = _get_embedding_model_class(cfg.embedding_model_name)(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=model, tokenizer=tok)
    assert em.encoder_path.startswith("transformers (") and em.batch_encode(["midnight"]).shape == (1, 128)
    em.close()
    assert fused_bert.lens_of_mask(np.array([[0, 1, 1], [1, 1, 1]])) is None and fused_bert.lens_of_mask(np.array([[1, 1, 0], [1, 0, 0]])).tolist() == [2, 1]


def test_captured_graphs_equal_eager_forwards_also_from_many_threads():
    """A mini-batch shape seen twice is replayed as a captured hipGraph with static buffers: same rows as the eager pass,
    for interleaved shapes and for 8 threads sharing one model (ComoRAG.py:436-441 runs up to 16 over one instance)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    model, tok = _peaked_tiny_bert(torch.bfloat16)
    cls = _get_embedding_model_class("bge-tiny-random")
    mk = lambda **kw: cls(global_config=BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128,
                                                   embedding_model_dtype="bf16", **kw),
                          embedding_model_name="bge-tiny-random", model=copy.deepcopy(model), tokenizer=tok)
    em, eager = mk(), mk(embedding_hip_graphs=0)
    queries = ["midnight", "what did the mother wish " * 3, "the prince and the golden slipper " * 6, "she was good and pious " * 12,
               "who how when", "the bird in the tree and the king and his son went to the dance " * 2]
    want = [eager.batch_encode(q) for q in queries]
    assert not eager._fused._graphs
    for rep in range(4):                                   # rep 0 eager, rep 1 captures, reps 2-3 replay
        for q, w in zip(queries, want):
            np.testing.assert_allclose(em.batch_encode(q), w, atol=2e-6)
    shapes = set(em._fused._graphs)
    assert 1 <= len(shapes) <= len(queries) and all(l % 16 == 0 for _, l, _ in shapes)
    batch = em.batch_encode(queries)                        # a 6-row mini-batch: another shape, first eager
    for _ in range(3):
        np.testing.assert_allclose(em.batch_encode(queries), batch, atol=2e-6)
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda i: em.batch_encode(queries[i % len(queries)]), range(96)))
    for i, g in enumerate(got):
        np.testing.assert_allclose(g, want[i % len(queries)], atol=2e-6)
    em.close(); eager.close()
