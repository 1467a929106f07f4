"""leaf_pytorch_amd -- MI355X-native LEAF frontend (hand-written HIP kernels for gfx950) behind the
``leaf_pytorch.frontend.Leaf`` module surface of SarthakYadav/leaf-pytorch."""
from .frontend import Leaf, SquaredModulus
from .frontend_helper import get_frontend
from .modules import (ExponentialMovingAverage, GaborConstraint, GaborConv1d, GaussianLowPass, PCENLayer,
                      get_padding_value)
from .initializers import GaborFilter, GaborInit
from ._native import build, load, LIB_PATH
from .transforms import CenterCrop, PeakNormalization, RandomCrop
from .streaming import LeafStream

__all__ = ["Leaf", "SquaredModulus", "get_frontend", "GaborConv1d", "GaborConstraint", "GaussianLowPass",
           "ExponentialMovingAverage", "PCENLayer", "GaborInit", "GaborFilter", "get_padding_value", "build", "load",
           "PeakNormalization", "CenterCrop", "RandomCrop", "LeafStream"]
