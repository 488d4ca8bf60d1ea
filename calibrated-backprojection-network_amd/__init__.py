"""MI355X-native KBNet inference hot path (S2D pool, KB layers, encoder-decoder).

The directory name contains hyphens, so import it with
`importlib.import_module("calibrated-backprojection-network_amd")` or through the
root-level alias module `kbnet_amd`.
"""

from .config import KBNetConfig, kitti_config, void_config, PRESETS  # noqa: F401

__version__ = "0.1.0"
