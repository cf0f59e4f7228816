"""mednet_hip: MI355X-native 3D U-Net forward/backward + per-voxel losses behind the nn.Module surface of
midasmednet.unet (tobiashepp/torch-mednet).  See DESIGN.md."""
from . import config
from .config import get_precision, precision, set_conv_algo, set_precision  # noqa: F401

__all__ = ["config", "set_precision", "get_precision", "precision", "set_conv_algo"]
