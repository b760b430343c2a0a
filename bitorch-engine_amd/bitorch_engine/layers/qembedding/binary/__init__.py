from .layer import BinaryEmbeddingBag, BinaryEmbeddingParameter, BinaryEmbeddingCuda  # noqa: F401
