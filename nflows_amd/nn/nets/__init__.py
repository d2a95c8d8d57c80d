from .resnet import ResidualBlock, ResidualNet
from .mlp import MLP
