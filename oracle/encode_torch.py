"""CPU restatement (torch fp32) of BGEEmbeddingModel._encode / batch_encode.

TEST INFRASTRUCTURE ONLY (see oracle/retrieval_np.py header).  Follows
src/comorag/embedding_model/BGEEmbedding.py:92-129 (prefix concat with no separator, HF tokenizer
padding/truncation, AutoModel forward, mean_pooling :15-28, F.normalize) and :131-185 (fixed
instruction overwrite, mini-batch loop, cat, numpy).  Pinned against the live reference class in
tests/test_oracle_pin.py when /root/reference exists.  BGE weights / vocab are not on disk and there
is no network: `tiny_bert()` builds the seed-initialised BertModel + synthetic WordPiece tokenizer
both the oracle and the HIP path are fed with.
"""
from __future__ import annotations

from typing import List

import numpy as np

BGE_PREFIX = "Generate a representation for this sentence to retrieve relevant articles:"


def tiny_bert(hidden=64, layers=2, heads=4, inter=128, max_pos=128, vocab_words=None, seed=0):
    import torch
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import BertConfig, BertModel, PreTrainedTokenizerFast
    words = vocab_words or ("the a an and of to in she he it her his was were had be good pious mother grave snow spring prince "
                            "slipper golden ball pumpkin coach midnight stepmother sisters bird tree wish dress dance king son "
                            "generate representation for this sentence retrieve relevant articles what who how did when").split()
    vocab = {t: i for i, t in enumerate(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])}
    for w in words + [f"##{c}" for c in "abcdefghijklmnopqrstuvwxyz"] + list("abcdefghijklmnopqrstuvwxyz") + list(".,:;?!'\"-"):
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordPiece(vocab=vocab, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1",
                                                       special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]",
                                   sep_token="[SEP]", mask_token="[MASK]")
    torch.manual_seed(seed)
    cfg = BertConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=inter, max_position_embeddings=max_pos)
    model = BertModel(cfg, add_pooling_layer=False).eval()
    return model, fast


def tiny_xlmr(hidden=256, layers=2, heads=4, inter=512, max_pos=2050, seed=0):
    """A seed-initialised XLM-RoBERTa encoder (bge-m3's architecture: position ids start at padding_idx + 1 = 2, one token type,
    LayerNorm eps 1e-5) + a synthetic tokenizer whose <pad> has id 1 like XLM-R's (no sentencepiece model exists offline)."""
    import torch
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast, XLMRobertaConfig, XLMRobertaModel
    words = ("the a an and of to in she he it her his was were had be good pious mother grave snow spring prince slipper golden ball "
             "pumpkin coach midnight stepmother sisters bird tree wish dress dance king son generate representation for this sentence "
             "retrieve relevant articles what who how did when").split()
    vocab = {t: i for i, t in enumerate(["<s>", "<pad>", "</s>", "<unk>", "<mask>"])}
    for w in words + [f"##{c}" for c in "abcdefghijklmnopqrstuvwxyz"] + list("abcdefghijklmnopqrstuvwxyz") + list(".,:;?!'\"-"):
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordPiece(vocab=vocab, unk_token="<unk>"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", unk_token="<unk>", cls_token="<s>", sep_token="</s>",
                                   bos_token="<s>", eos_token="</s>", mask_token="<mask>")
    torch.manual_seed(seed)
    cfg = XLMRobertaConfig(vocab_size=len(vocab), hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter,
                           max_position_embeddings=max_pos, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2, layer_norm_eps=1e-5)
    model = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    return model, fast


def mean_pooling(token_embeddings, mask):
    """BGEEmbedding.py:15-28."""
    token_embeddings = token_embeddings.masked_fill(~mask[..., None].bool(), 0.)
    return token_embeddings.sum(dim=1) / mask.sum(dim=1)[..., None]


def encode(model, tokenizer, prompts: List[str], instruction: str = "", max_length: int = 512, normalize: bool = True):
    """BGEEmbedding.py:92-129 on the model's own device, fp32."""
    import torch
    if isinstance(prompts, str):
        prompts = [prompts]
    if instruction:
        prompts = [instruction + t for t in prompts]
    with torch.no_grad():
        dev = next(model.parameters()).device
        inputs = tokenizer(prompts, padding=True, truncation=True, max_length=max_length, return_tensors="pt").to(dev)
        out = model(**inputs)
        emb = mean_pooling(out.last_hidden_state, inputs["attention_mask"])
        if normalize:
            emb = torch.nn.functional.normalize(emb, p=2, dim=1)
    return emb


def batch_encode(model, tokenizer, texts, batch_size: int = 32, max_length: int = 512) -> np.ndarray:
    """BGEEmbedding.py:131-185 with the kwargs the callers actually produce (instruction is always
    overwritten with the fixed prefix)."""
    import torch
    if isinstance(texts, str):
        texts = [texts]
    parts = [encode(model, tokenizer, texts[i:i + batch_size], instruction=BGE_PREFIX, max_length=max_length)
             for i in range(0, len(texts), batch_size)]
    return torch.cat(parts, dim=0).cpu().numpy()
