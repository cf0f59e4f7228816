from mednet_hip.unet.loss import *  # noqa: F401,F403
from mednet_hip.unet.loss import (flatten, compute_per_channel_dice, dice_metric, expand_as_one_hot, DiceLoss, CELoss,  # noqa: F401
                                  WeightedCrossEntropyLoss, BCELossWrapper, PixelWiseCrossEntropyLoss, LandmarkLoss)
