"""minivectordb_amd — MI355X-native drop-in for MiniVectorDB's embed+search hot path.

Public surface mirrors the reference package (minivectordb/): ``VectorDatabase``,
``ShardedVectorDatabase``, ``EmbeddingModel`` / ``AlternativeModel``.  All numeric work runs in
hand-written HIP kernels behind the C-ABI in include/mvdb.h; there is no CPU fallback.
"""

import os as _os

# RCCL / cross-process device tensors need dmabuf IPC on this host driver; it has to be in the environment before the
# first HIP call of the process (harmless for single-GPU use)
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__all__ = ["VectorDatabase", "ShardedVectorDatabase", "DistributedShardedVectorDatabase", "EmbeddingModel",
           "AlternativeModel"]


def __getattr__(name):  # lazy: importing the package must not require a GPU
    if name == "VectorDatabase":
        from .vector_database import VectorDatabase
        return VectorDatabase
    if name == "ShardedVectorDatabase":
        from .sharded_vector_database import ShardedVectorDatabase
        return ShardedVectorDatabase
    if name == "DistributedShardedVectorDatabase":
        from .distributed import DistributedShardedVectorDatabase
        return DistributedShardedVectorDatabase
    if name in ("EmbeddingModel", "AlternativeModel"):
        from . import embedding_model
        return getattr(embedding_model, name)
    raise AttributeError(name)
