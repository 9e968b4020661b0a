"""kinetic-gan_amd: MI355X-native st_gcn hot path of Kinetic-GAN."""
from . import graph  # noqa: F401

__version__ = "0.1.0"
