from .shards import Metadata, ShardInfo, write_shards
from .shuffled import Config as ShuffledConfig
from .shuffled import DataLoader as ShuffledDataLoader

__all__ = ["Metadata", "ShardInfo", "ShuffledConfig", "ShuffledDataLoader", "write_shards"]
