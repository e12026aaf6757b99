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
        self.add_layernorm_pool = FusedBertLayers.add_layernorm_pool.__get__(self)


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
@pytest.mark.parametrize("d,rows", [(768, 1001), (1024, 64), (256, 7), (40, 130), (2048, 33), (1536, 5), (1024, 600), (2048, 300), (256, 257), (768, 256)])
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
@pytest.mark.parametrize("b,l,d", [(5, 64, 768), (32, 512, 768), (3, 128, 1024), (2, 16, 256), (4, 48, 2048), (6, 48, 2048), (1, 32, 72)])
@pytest.mark.parametrize("normalize", [True, False])
def test_last_layer_layernorm_with_the_encoder_tail_folded_in(dtype, b, l, d, normalize):
    """cmr_encoder_add_layernorm_pool == cmr_encoder_add_layernorm followed by cmr_pool_l2norm (mean_pooling + F.normalize,
    BGEEmbedding.py:15-28, :126-127) on the same inputs — ragged right-padded lengths incl. a one-token row, a row that ends
    inside a 16-token block and a full one — and both against the fp32 formula on the 16-bit-rounded LayerNorm output."""
    import torch
    from comorag_amd.embedding_model.bge import pool_l2norm
    dt = getattr(torch, dtype)
    g = torch.Generator(device="cuda"); g.manual_seed(b * 1000 + l + d)
    y = torch.randn((b * l, d), generator=g, device="cuda").to(dt)
    res = torch.randn((b * l, d), generator=g, device="cuda").to(dt)
    bias, gamma, beta = (torch.randn(d, generator=g, device="cuda").to(dt) for _ in range(3))
    lens = torch.tensor(([1, l, max(1, l - 5), max(1, l // 2 + 3), 17] * 8)[:b], dtype=torch.int32).clamp_(max=l)
    mask = (torch.arange(l)[None, :] < lens[:, None]).to(torch.int64)
    fz = _Stages(d, 1, dt, eps=1e-12)
    hidden = fz.add_layernorm(y, bias, res, gamma, beta).view(b, l, d)
    two = pool_l2norm(hidden, mask.cuda(), normalize=normalize)
    one = fz.add_layernorm_pool(y, bias, res, gamma, beta, lens.cuda(), b, l, normalize)
    torch.cuda.synchronize()
    want = (hidden.float() * mask.cuda()[..., None]).sum(1) / lens.cuda()[:, None].float()
    if normalize:
        want = torch.nn.functional.normalize(want, p=2, dim=1)
    tol = 3e-6 if normalize else 3e-5
    if b * l <= 256:
        # A handful of rows: cmr_encoder_add_layernorm runs its wave-per-row kernel (round 6: one short query's 24 / 48 LayerNorm launches at
        # 5-7 us instead of 9-12), whose fp32 sums run in another order than the folded kernel's sixteen-lane groups: a LayerNorm output now
        # and then rounds to the neighbouring 16-bit value, and a one-token row shows that unit in the last place undiluted.
        ulp = 2.0 ** -8 if dtype == "bfloat16" else 2.0 ** -11
        tol = max(tol, 1.5 * ulp * float(want.abs().max()))
    np.testing.assert_allclose(one.cpu().numpy(), want.cpu().numpy(), atol=tol, rtol=1e-5)
    np.testing.assert_allclose(one.cpu().numpy(), two.cpu().numpy(), atol=tol, rtol=1e-5)


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


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_gelu_in_the_gemm_epilogue_vs_the_exact_kernel_vs_oracle(dtype):
    """`embedding_gelu = "epilogue"` (opt-in): FFN-up GEMM + bias + GELU as ONE hipBLASLt launch, GELU in its tanh form;
    `"exact"` (default): the GEMM, then PyTorch's erf-form kernel (what transformers runs).  Both stacks are held to the SAME bar against the fp32
    oracle (erf form, BGEEmbedding.py:119), and to each other within the 16-bit rounding of the activations they differ in."""
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.embedding_model.fused_bert import gelu_epilogue_available
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = _peaked_tiny_bert(getattr(torch, dtype))
    with torch.no_grad():
        for lyr in model.encoder.layer:                 # pre-activations of a few units: where the two GELU forms differ most (|x| ~ 2.7)
            lyr.intermediate.dense.weight.mul_(6.0)
        model.to(getattr(torch, dtype)).float()
    texts = [f"the prince and the golden slipper number {i} " + "and the bird in the tree " * (i % 7) for i in range(19)] + ["midnight"]
    want = enc.batch_encode(model, tok, texts, batch_size=8, max_length=128)
    cls = _get_embedding_model_class("bge-tiny-random")
    out = {}
    for mode in ("epilogue", "exact"):
        cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128, embedding_model_dtype=dtype, embedding_gelu=mode)
        em = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
        assert em.encoder_path == "hip-fused-layers"
        out[mode] = (em.batch_encode(texts), em._fused.gelu_path)
        em.close()
    assert out["exact"][1] == "exact-erf-kernel"
    assert out["epilogue"][1] == ("hipblaslt-epilogue-tanh" if gelu_epilogue_available(torch.device("cuda", 0), getattr(torch, dtype)) else "exact-erf-kernel")
    tol = {"bfloat16": 6e-3, "float16": 1e-3}[dtype]
    for mode in out:
        got = out[mode][0]
        np.testing.assert_allclose(got, want, atol=tol)
        assert float(np.min((got * want).sum(1))) > 0.999          # north_star: cosine within 1e-3 of the reference path
        assert np.abs(got @ got.T - want @ want.T).max() < 1e-3     # every pairwise score within 1e-3
    assert np.abs(out["epilogue"][0] - out["exact"][0]).max() <= tol


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_both_gelu_paths_on_heavy_tailed_pre_activations_at_bert_base_depth(dtype):
    """The exactness gate of the opt-in epilogue path.  Real BGE weights are not in the image, and seed-initialised
    BERT pre-activations are near-Gaussian with |x| < 4 — so the FFN-up rows of a 12-layer BERT-base shape get log-normal scales
    (a scale mixture over the units) until the pre-activations of EVERY layer are heavy-tailed (kurtosis >> 3, |x| beyond 8,
    a good share of them in the 2 <= |x| <= 3.5 band where the tanh and erf forms of GELU differ most).  Bars against the fp32 oracle
    (erf form, BGEEmbedding.py:119-120): fp16 — min row cosine >= 0.9995, every pairwise score within 5e-4, both paths; bf16 — activations
    of magnitude 30-45 carry 8 mantissa bits, the EXACT path itself sits at 0.9992 / 1.9e-3 here (measured), so the bar is the dtype's:
    0.999 / 3e-3; and in both dtypes the epilogue path may be no further from the oracle than the exact path by more than 2e-4 in cosine /
    3e-4 in any pairwise score, the two paths' rows within 0.9995 of each other.  The default must be the exact one."""
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    assert BaseConfig().embedding_gelu == "exact"
    tdt = getattr(torch, dtype)
    model, tok = enc.tiny_bert(hidden=768, layers=12, heads=12, inter=3072, max_pos=128)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for lyr in model.encoder.layer:
            lyr.attention.self.query.weight.mul_(8.0)
            lyr.attention.self.key.weight.mul_(8.0)
            w = lyr.intermediate.dense.weight
            # a scale mixture over the units: unit j's pre-activation is ~N(0, s_j^2), s_j log-normal (a sum over 768 inputs is Gaussian
            # per unit whatever the weights' law: the heavy tail has to come from the units' scales) — kurtosis ~ 3 exp(4 * 0.7^2) ~ 20
            scale = torch.exp(0.7 * torch.randn((w.shape[0], 1), generator=g))
            w.copy_(torch.randn(w.shape, generator=g) * scale * (1.2 / 768 ** 0.5))
            lyr.intermediate.dense.bias.copy_(torch.randn(w.shape[0], generator=g) * 0.5)
        model.to(tdt).float()
    texts = [f"the prince and the golden slipper number {i} " + "and the bird in the tree " * (i % 9) for i in range(15)] + ["midnight"]
    pre = []
    hooks = [lyr.intermediate.dense.register_forward_hook(lambda m, i, o: pre.append(o.detach().flatten())) for lyr in model.encoder.layer]
    want = enc.batch_encode(model, tok, texts, batch_size=8, max_length=128)
    for h in hooks:
        h.remove()
    for x in pre:                                                              # the regime, layer by layer (padding rows included: same weights)
        x = x.double()
        kurt = float(((x - x.mean()) ** 4).mean() / x.var() ** 2)
        band = float(((x.abs() >= 2.0) & (x.abs() <= 3.5)).double().mean())
        assert kurt > 6.0 and float(x.abs().max()) >= 8.0 and band > 0.05, (kurt, float(x.abs().max()), band)
    cls = _get_embedding_model_class("bge-tiny-random")
    rep, rows = {}, {}
    for mode in ("exact", "epilogue"):
        cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128, embedding_model_dtype=dtype, embedding_gelu=mode)
        em = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
        assert em.encoder_path == "hip-fused-layers"
        got = em.batch_encode(texts).astype(np.float64)
        path = em._fused.gelu_path
        em.close()
        w64 = want.astype(np.float64)
        rep[mode] = (path, float((got * w64).sum(1).min()), float(np.abs(got @ got.T - w64 @ w64.T).max()))
        rows[mode] = got
    print("heavy-tailed GELU gate", dtype, rep)
    assert rep["exact"][0] == "exact-erf-kernel"
    cos_bar, pair_bar = {"bfloat16": (0.999, 3e-3), "float16": (0.9995, 5e-4)}[dtype]
    for mode, (path, cos, pair) in rep.items():
        assert cos >= cos_bar and pair <= pair_bar, (mode, path, cos, pair)
    assert rep["epilogue"][1] >= rep["exact"][1] - 2e-4 and rep["epilogue"][2] <= rep["exact"][2] + 3e-4, rep
    assert float((rows["epilogue"] * rows["exact"]).sum(1).min()) >= cos_bar


def test_fused_forward_launches_no_gelu_kernel():
    """With the epilogue path the encoder's forward holds NO GELU launch of its own (the activation is inside the FFN-up GEMM): the kernel
    list of one fused forward, taken with torch's profiler, names no Gelu kernel — the exact path's list does."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    from comorag_amd.embedding_model.fused_bert import FusedBertLayers
    model, _ = _peaked_tiny_bert(torch.bfloat16)
    model = model.to("cuda", dtype=torch.bfloat16).eval()
    ids = torch.randint(0, model.config.vocab_size, (4, 64), device="cuda")
    lens = np.array([64, 40, 17, 64], np.int32)
    names = {}
    for mode in ("epilogue", "exact"):
        fz = FusedBertLayers(model, gelu=mode)
        fz(ids, lens, pool=True); torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fz(ids, lens, pool=True); torch.cuda.synchronize()
        names[mode] = (fz.gelu_path, [e.key for e in prof.key_averages()])
    assert names["exact"][0] == "exact-erf-kernel" and any("elu" in n for n in names["exact"][1]), names["exact"][1]
    if names["epilogue"][0] == "hipblaslt-epilogue-tanh":
        assert not any("Gelu" in n or "gelu" in n for n in names["epilogue"][1]), names["epilogue"][1]


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
    em, eager = mk(embedding_query_cache=0), mk(embedding_hip_graphs=0, embedding_query_cache=0)      # (every call below is a forward, not a cached row)
    queries = ["midnight", "what did the mother wish " * 3, "the prince and the golden slipper " * 6, "she was good and pious " * 12,
               "who how when", "the bird in the tree and the king and his son went to the dance " * 2]
    want = [eager.batch_encode(q) for q in queries]
    assert not eager._fused._graphs
    for rep in range(4):                                   # rep 0 eager, rep 1 captures, reps 2-3 replay
        for q, w in zip(queries, want):
            np.testing.assert_allclose(em.batch_encode(q), w, atol=2e-6)
    shapes = set(em._fused._graphs)
    assert 1 <= len(shapes) <= len(queries) and all(key[1] % 16 == 0 for key in shapes)
    batch = em.batch_encode(queries)                        # a 6-row mini-batch: another shape, first eager
    for _ in range(3):
        np.testing.assert_allclose(em.batch_encode(queries), batch, atol=2e-6)
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(lambda i: em.batch_encode(queries[i % len(queries)]), range(96)))
    for i, g in enumerate(got):
        np.testing.assert_allclose(g, want[i % len(queries)], atol=2e-6)
    em.close(); eager.close()


def test_single_string_results_are_kept_and_handed_back_as_copies():
    """ComoRAG encodes a question three times per tri_retrieve (ComoRAG.py:921-935 once per instruction — which its BGE model ignores — and
    get_similar_summaries once more): HipBGEEmbeddingModel keeps the last `embedding_query_cache` single-string results by (prompt, max_length,
    normalisation).  A hit is the first call's row, a COPY of it (the caller may scribble on it), keyed on what changes the row; bounded;
    lists of strings and device-resident results never look there; 0 switches it off."""
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    model, tok = _peaked_tiny_bert(torch.bfloat16)
    cls = _get_embedding_model_class("bge-tiny-random")
    mk = lambda **kw: cls(global_config=BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128,
                                                   embedding_model_dtype="bf16", **kw),
                          embedding_model_name="bge-tiny-random", model=copy.deepcopy(model), tokenizer=tok)
    em, off = mk(embedding_query_cache=3), mk(embedding_query_cache=0)
    forwards = []
    real = em._encode
    em._encode = em.encode = lambda *a, **k: (forwards.append(1), real(*a, **k))[1]
    a = em.batch_encode("who lost a slipper", instruction="query_to_fact", norm=True)
    b = em.batch_encode("who lost a slipper", instruction="query_to_passage", norm=True)          # the instruction is overwritten by the fixed prefix: same prompt
    assert len(forwards) == 1 and np.array_equal(a, b) and a is not b
    np.testing.assert_array_equal(a, off.batch_encode("who lost a slipper"))
    a[:] = 7.0                                                                                      # a caller's scribble stays the caller's
    assert np.array_equal(em.batch_encode("who lost a slipper"), b) and len(forwards) == 1
    em.batch_encode("who lost a slipper", max_length=8); em.batch_encode("who lost a slipper", normalize=False)
    assert len(forwards) == 3                                                                       # other rows: other keys
    em.batch_encode(["who lost a slipper"] * 2); em.batch_encode_dev("who lost a slipper")
    assert len(forwards) == 5                                                                       # lists and device results are not cached
    for q in ("one", "two", "three"):
        em.batch_encode(q)
    n = len(forwards)
    em.batch_encode("who lost a slipper")                                                           # evicted by three newer keys
    assert len(forwards) == n + 1 and len(em._qcache) == 3
    off.batch_encode("midnight"); off.batch_encode("midnight")
    assert len(off._qcache) == 0
    em.close(); off.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind,dtype", [("base", "bf16"), ("large", "fp16")])
def test_bge_base_and_large_shapes_ragged_512_tokens_vs_fp32_oracle(kind, dtype):
    """The shapes BASELINE names — BGE-base (12 x 768, bf16) and BGE-large (24 x 1024, fp16) — through the product path
    (tokenise -> fused 16-bit layer stack -> HIP pool + L2-norm) on 32 ragged chunks of up to 512 tokens, against the oracle's
    fp32 restatement of the reference's batch_encode on the same (16-bit-representable) weights: north_star's
    'cosine scores within 1e-3 for bf16' per row and per pairwise score.  Random-init attention is near-uniform, so the
    query / key projections are scaled up to make the softmax matter (as in the toy-size test above)."""
    import torch
    from comorag_amd.embedding_model.bge import HipBGEEmbeddingModel
    from comorag_amd.utils.config_utils import BaseConfig
    from tools.bench_extras import encoder_parity
    from tools.synthetic import random_bert, synthetic_chunks, synthetic_wordpiece_tokenizer
    tok, words = synthetic_wordpiece_tokenizer()
    model = random_bert(kind, vocab_size=len(tok))
    with torch.no_grad():
        for lyr in model.encoder.layer:
            lyr.attention.self.query.weight.mul_(2.0)          # logit std ~1.5 (x6 saturates the softmax: an arg-max flip per rounding error,
            lyr.attention.self.key.weight.mul_(2.0)            # the fp32 / 16-bit comparison then measures chaos, not arithmetic)
    cfg = BaseConfig(embedding_model_name=f"bge-{kind}-random-init", embedding_batch_size=8, embedding_model_dtype=dtype)
    em = HipBGEEmbeddingModel(cfg, cfg.embedding_model_name, model=model, tokenizer=tok)
    assert em.encoder_path == "hip-fused-layers"
    chunks = synthetic_chunks(words, 32, tokens_per_chunk=560)
    r = encoder_parity(torch, em, chunks)
    assert r["min_row_cosine_vs_fp32_oracle"] >= 1.0 - 1e-3, r
    assert r["max_abs_pairwise_score_diff"] <= 1e-3, r
    # and no further from fp32 than the reference's own code run in the same 16-bit dtype (transformers forward + torch pooling)
    assert 1.0 - r["min_row_cosine_vs_fp32_oracle"] <= 2.0 * (1.0 - r["transformers_same_dtype_min_row_cosine"]) + 1e-5, r
    em.close()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_xlm_roberta_stack_at_2048_tokens_vs_fp32_oracle(dtype):
    """bge-m3's architecture (XLM-RoBERTa: position ids from padding_idx + 1, one token type, eps 1e-5) through the fused stack at
    the reference's default embedding_max_seq_len = 2048 (utils/config_utils.py:140) — the one BGE model that length is safe with
    (SURVEY.md 5): a row that fills all 2048 positions, ragged ones, a two-token one, against the fp32 oracle (the reference's
    batch_encode restated, HF's own position ids) on the same 16-bit-representable weights, and the transformers forward."""
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.embedding_model import fused_bert
    from comorag_amd.utils.config_utils import BaseConfig
    from oracle import encode_torch as enc
    model, tok = enc.tiny_xlmr(hidden=256, layers=2, heads=4, inter=512, max_pos=2050)
    with torch.no_grad():
        for lyr in model.encoder.layer:
            lyr.attention.self.query.weight.mul_(10.0)
            lyr.attention.self.key.weight.mul_(10.0)
        model.to(getattr(torch, dtype)).float()
    assert fused_bert.position_offset(model) == 2 and fused_bert.why_not(copy.deepcopy(model).to(torch.bfloat16)) is None
    texts = ["the prince and the golden slipper and the bird in the tree " * 200, "she was good and pious " * 150, "midnight",
             "what did the mother wish " * 60, "who how when " * 11]
    want = enc.batch_encode(model, tok, texts, batch_size=4, max_length=2048)
    cls = _get_embedding_model_class("bge-m3-random")
    out = {}
    for fused in (True, False):
        cfg = BaseConfig(embedding_model_name="bge-m3-random", embedding_batch_size=4, embedding_model_dtype=dtype, embedding_fused_encoder=fused)
        assert cfg.embedding_max_seq_len == 2048
        em = cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
        assert em.max_positions == 2048
        assert em.encoder_path == ("hip-fused-layers" if fused else "transformers")
        out[fused] = em.batch_encode(texts)
        em.close()
    ntok = [len(tok(enc.BGE_PREFIX + t, truncation=True, max_length=2048)["input_ids"]) for t in texts]
    assert max(ntok) == 2048 and min(ntok) < 40
    tol = {"bfloat16": 6e-3, "float16": 1e-3}[dtype]
    np.testing.assert_allclose(out[True], want, atol=tol)
    np.testing.assert_allclose(out[False], want, atol=tol)
    assert np.abs(out[True] - want).max() <= 2.0 * np.abs(out[False] - want).max() + 2e-4
    assert float(np.min((out[True] * want).sum(1))) > 0.9995


@pytest.mark.parametrize("kind", ["bert", "xlmr"])
def test_ragged_mini_batches_equal_padded_ones(kind):
    """The host ships ONE int32 array lens | offsets | ids per mini-batch (FusedBertLayers.forward_ragged, embed_ln_ragged_kernel)
    instead of the padded id / mask / token-type tensors of BGEEmbedding.py:112-118: same embeddings rows for the real tokens,
    same pooled output as the padded call, eager and as a captured graph."""
    import torch
    from comorag_amd.embedding_model.fused_bert import FusedBertLayers
    from oracle import encode_torch as enc
    model, tok = (enc.tiny_bert(hidden=256, layers=2, heads=4, inter=512, max_pos=160) if kind == "bert"
                  else enc.tiny_xlmr(hidden=256, layers=2, heads=4, inter=512, max_pos=162))
    fz = FusedBertLayers(model.to("cuda", dtype=torch.bfloat16), graphs=4)
    rng = np.random.default_rng(7)
    lens = np.array([1, 160, 33, 16, 97, 2], np.int32)
    ids = [rng.integers(5, len(tok), n).astype(np.int32) for n in lens]
    b, l = len(lens), 160
    padded = np.full((b, l), tok.pad_token_id, np.int64)
    for r, x in enumerate(ids):
        padded[r, :len(x)] = x
    head = np.concatenate([lens, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)])
    packed = np.concatenate([head, *ids]).astype(np.int32)
    pd = torch.from_numpy(packed).cuda()
    a = fz.embed(torch.from_numpy(padded).cuda(), None).view(b, l, -1)
    r = fz.embed_ragged(pd[2 * b + 1:], pd[b:2 * b + 1], b, l).view(b, l, -1)
    for i, n in enumerate(lens):
        assert torch.equal(a[i, :n], r[i, :n])
    want = fz(torch.from_numpy(padded).cuda(), lens, pool=True).cpu().numpy()
    for rep in range(3):                                           # eager, capture, replay
        got = fz.forward_ragged(packed, b, l, True).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    assert any(k[2] == "ragged" for k in fz._graphs)
    # more shapes than the table holds: the least recently replayed graph goes, the newest shape is captured
    for width in (16, 32, 48, 64, 80):
        short = np.concatenate([[3, 5], [0, 3, 8], rng.integers(5, len(tok), 8)]).astype(np.int32)
        for rep in range(3):
            fz.forward_ragged(short, 2, width, True)
    assert len(fz._graphs) == 4 and (2, 80, "ragged", True) in fz._graphs and (2, 16, "ragged", True) not in fz._graphs
    fz.release()


@pytest.mark.parametrize("replicas", [2, 4])
def test_corpus_encode_over_replicas_equals_the_single_replica_rows(replicas):
    """`embedding_devices` / `embedding_encode_replicas` (the reference's `device_map="auto"  # Use multiple GPUs if available`,
    BGEEmbedding.py:77): one copy of the fused layer stack per replica, each on its own stream and worker thread, the bucketing windows
    of a corpus-sized batch_encode dealt round them, rows gathered device to device in arrival order.  Rehearsal on the one GPU of a
    test box: R logical replicas on cuda:0 must return the rows of the single-replica path BIT FOR BIT (same windows, same mini-batches,
    same kernels), through batch_encode and batch_encode_dev, twice (captured graphs on the second pass), and a query-sized call must not
    touch the replicas."""
    import torch
    from comorag_amd.embedding_model import _get_embedding_model_class
    from comorag_amd.utils.config_utils import BaseConfig
    model, tok = _peaked_tiny_bert(torch.bfloat16)
    texts = [f"the prince and the golden slipper number {i} " + "and the bird in the tree " * (i % 11) + "midnight " * (i % 3) for i in range(203)]
    cls = _get_embedding_model_class("bge-tiny-random")
    def make(**kw):
        cfg = BaseConfig(embedding_model_name="bge-tiny-random", embedding_batch_size=8, embedding_max_seq_len=128, embedding_model_dtype="bfloat16", **kw)
        return cls(global_config=cfg, embedding_model_name=cfg.embedding_model_name, model=copy.deepcopy(model), tokenizer=tok)
    one = make()
    want = one.batch_encode(texts)
    many = make(embedding_devices=[0], embedding_encode_replicas=replicas)
    assert len(one._replicas) == 0 and len(many._replicas) == replicas and many._replicas[0].fused is many._fused
    assert len({id(r.fused) for r in many._replicas}) == replicas and len({r.stream.cuda_stream for r in many._replicas}) == replicas
    for rep in range(2):
        got = many.batch_encode(texts)
        assert got.shape == want.shape and np.array_equal(got, want), (rep, float(np.abs(got - want).max()))
    dev = many.batch_encode_dev(texts[:77])
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), one.batch_encode(texts[:77]))
    # every replica did work; a single query stays on the owner's stack
    assert all(len(r.fused._seen) + len(r.fused._graphs) > 0 for r in many._replicas)
    seen = [dict(r.fused._seen) for r in many._replicas[1:]]
    np.testing.assert_array_equal(many.batch_encode("midnight"), one.batch_encode("midnight"))
    assert seen == [dict(r.fused._seen) for r in many._replicas[1:]]
    with pytest.raises(ValueError):
        make(embedding_devices=[torch.cuda.device_count()])
    one.close(); many.close()
    assert many._replicas == []
