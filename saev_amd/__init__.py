"""saev_amd — MI355X-native TopK sparse-autoencoder train step behind saev's module API."""

__version__ = "0.1.0"
