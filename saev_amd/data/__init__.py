from .extract import ActivationRecorder, DeviceReservoir, ExtractConfig, ExtractionFeed
from .ordered import Config as OrderedConfig
from .ordered import DataLoader as OrderedDataLoader
from .shards import Metadata, ShardInfo, write_shards
from .shuffled import Config as ShuffledConfig
from .shuffled import DataLoader as ShuffledDataLoader

__all__ = ["ActivationRecorder", "DeviceReservoir", "ExtractConfig", "ExtractionFeed", "Metadata", "OrderedConfig", "OrderedDataLoader", "ShardInfo", "ShuffledConfig", "ShuffledDataLoader", "write_shards"]
