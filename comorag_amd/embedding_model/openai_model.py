"""OpenAIEmbeddingModel — remote HTTP embeddings (src/comorag/embedding_model/OpenAI.py:77-128).
Not on the GPU path; kept so the factory covers the same names.  Needs the `openai` package."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .base import BaseEmbeddingModel, EmbeddingConfig


class OpenAIEmbeddingModel(BaseEmbeddingModel):
    def __init__(self, global_config=None, embedding_model_name: Optional[str] = None, client=None) -> None:
        super().__init__(global_config=global_config)
        if embedding_model_name is not None:
            self.embedding_model_name = embedding_model_name
        self.embedding_config = EmbeddingConfig.from_dict({"embedding_model_name": self.embedding_model_name,
                                                           "encode_params": {"batch_size": 16}})
        if client is None:
            from openai import OpenAI  # ImportError here is the loud failure
            client = OpenAI()
        self.client = client
        self.embedding_dim = 1536

    def encode(self, texts: List[str]):
        texts = [t.replace("\n", " ") or " " for t in texts]
        resp = self.client.embeddings.create(input=texts, model=self.embedding_model_name)
        return np.array([v.embedding for v in resp.data])   # float64, as OpenAI.py:83

    def batch_encode(self, texts, **kwargs) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        bs = self.embedding_config.encode_params.get("batch_size", 16)
        out = [self.encode(texts[i:i + bs]) for i in range(0, len(texts), bs)]
        res = np.concatenate(out)
        if kwargs.get("norm", True):
            res = (res.T / np.linalg.norm(res, axis=1)).T
        return res
