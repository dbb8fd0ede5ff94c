"""Drop-in for the reference's `model` package (reference model/__init__.py): the same public names,
backed by libpesr_hip.so.  `from pesr_amd.model import *` (or the top-level `model` shim) gives
Generator, Discriminator, VGG, FocalLoss and, as the reference's star-import leaks them, nn / torch / F."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .basic import BasicBlock, Conv, MeanShift, PixelShuffle, ResBlock, Upsampler
from .focal_loss import FocalLoss
from .pesr import Discriminator, Generator
from .vgg import VGG

__all__ = ["Generator", "Discriminator", "VGG", "FocalLoss", "Conv", "MeanShift", "BasicBlock", "ResBlock",
           "Upsampler", "PixelShuffle", "nn", "torch", "F"]
