"""Factory with the reference's keys (src/comorag/embedding_model/__init__.py:10-17)."""
import logging

from .base import BaseEmbeddingModel, EmbeddingConfig, make_cache_embed  # noqa: F401

logger = logging.getLogger(__name__)


def _get_embedding_model_class(embedding_model_name: str = "None"):
    if "bge-" in embedding_model_name.lower():
        from .bge import HipBGEEmbeddingModel
        return HipBGEEmbeddingModel
    elif "text-embedding-3-small" in embedding_model_name:
        from .openai_model import OpenAIEmbeddingModel
        return OpenAIEmbeddingModel
    else:
        # the reference logs "using BGEEmbeddingModel as default" and then returns None (:15-17)
        logger.info(f"Unknown embedding model name: {embedding_model_name}")
        return None
