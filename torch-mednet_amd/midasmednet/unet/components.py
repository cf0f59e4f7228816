from mednet_hip.unet.components import *  # noqa: F401,F403
from mednet_hip.unet.components import (conv3d, create_conv, SingleConv, DoubleConv, ExtResNetBlock, Encoder, Decoder,  # noqa: F401
                                        FinalConv)
