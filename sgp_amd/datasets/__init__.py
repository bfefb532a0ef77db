from .iid_dataset import IIDSampler
from .sharded import ShardedEmbedding, ShardedIIDSampler

__all__ = ["IIDSampler", "ShardedEmbedding", "ShardedIIDSampler"]
