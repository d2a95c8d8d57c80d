from .resnet import ConvResidualBlock, ConvResidualNet, ResidualBlock, ResidualNet
from .mlp import MLP
