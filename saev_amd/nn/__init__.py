from .modeling import SparseAutoencoder, SparseAutoencoderConfig, dump, load
from .objectives import ObjectiveConfig, get_objective

__all__ = ["SparseAutoencoder", "SparseAutoencoderConfig", "ObjectiveConfig", "dump", "load", "get_objective"]
