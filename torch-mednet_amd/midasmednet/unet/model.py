from mednet_hip.unet.model import *  # noqa: F401,F403
from mednet_hip.unet.model import create_feature_maps, UNet3D, ResidualUNet3D  # noqa: F401
