"""Synthetic stand-ins for assets that are not on disk (no network): a random-init BERT encoder of
BGE-base / BGE-large shape and a WordPiece tokenizer over a generated vocabulary.  Bench / test data, not
part of the product package (comorag_amd/).  Used by bench.py
for the corpus-embed throughput figure; NOT a model — random weights give meaningless vectors."""
from __future__ import annotations


def synthetic_wordpiece_tokenizer(n_words: int = 30000, seed: int = 1234):
    import random
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    rnd = random.Random(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    vocab = {t: i for i, t in enumerate(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"])}
    for c in letters:
        vocab.setdefault(c, len(vocab)); vocab.setdefault("##" + c, len(vocab))
    words = []
    while len(vocab) < n_words:
        w = "".join(rnd.choice(letters) for _ in range(rnd.randint(2, 7)))
        if w not in vocab:
            vocab[w] = len(vocab); words.append(w)
    tok = Tokenizer(models.WordPiece(vocab=vocab, unk_token="[UNK]"))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]",
                                   sep_token="[SEP]", mask_token="[MASK]")
    return fast, words


def random_bert(kind: str = "base", vocab_size: int = 30000, seed: int = 0):
    import torch
    from transformers import BertConfig, BertModel
    shape = {"base": dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072),
             "large": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)}[kind]
    torch.manual_seed(seed)
    return BertModel(BertConfig(vocab_size=vocab_size, max_position_embeddings=512, **shape), add_pooling_layer=False).eval()


def synthetic_chunks(words, n_chunks: int, tokens_per_chunk: int = 480, seed: int = 1234):
    import random
    rnd = random.Random(seed)
    return [" ".join(rnd.choice(words) for _ in range(tokens_per_chunk)) for _ in range(n_chunks)]
