"""Builders for the BASELINE.json workloads, written against the
drop-in classes exactly as a user of the reference would write them."""
import torch

from .distributions import StandardNormal
from .flows import Flow
from .nn.nets import MLP, ResidualNet
from .transforms import (AffineCouplingTransform, CompositeTransform,
                         MaskedAffineAutoregressiveTransform,
                         MaskedPiecewiseRationalQuadraticAutoregressiveTransform,
                         PiecewiseRationalQuadraticCouplingTransform, RandomPermutation,
                         ReversePermutation)
from .utils import create_alternating_binary_mask


def rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, num_blocks=2,
                tail_bound=3.0, seed=0, activation=torch.nn.functional.relu):
    """configs[2] (16 layers) / configs[3] + north-star (32 layers): RandomPermutation +
    PiecewiseRationalQuadraticCouplingTransform(alternating mask, ResidualNet conditioner)."""
    if seed is not None:
        torch.manual_seed(seed)
    layers = []
    for i in range(num_layers):
        layers.append(RandomPermutation(features))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            mask=create_alternating_binary_mask(features, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: ResidualNet(
                i_, o_, hidden_features=hidden_features, num_blocks=num_blocks, activation=activation),
            num_bins=num_bins, tails="linear", tail_bound=tail_bound))
    return Flow(CompositeTransform(layers), StandardNormal([features]))


def affine_coupling_flow(num_layers=8, features=32, hidden_sizes=(128, 128), seed=0,
                         reverse_between=False):
    """configs[1]: AffineCouplingTransform stack with an MLP conditioner."""
    if seed is not None:
        torch.manual_seed(seed)
    layers = []
    for i in range(num_layers):
        layers.append(AffineCouplingTransform(
            mask=create_alternating_binary_mask(features, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: MLP([i_], [o_], list(hidden_sizes))))
        if reverse_between:
            layers.append(ReversePermutation(features))
    return Flow(CompositeTransform(layers), StandardNormal([features]))


def simple_realnvp_flow(features=16, hidden_features=128, num_layers=6, num_blocks_per_layer=2,
                        use_volume_preserving=False, seed=0):
    """The composition the reference's SimpleRealNVP factory builds (flows/realnvp.py:17-71; the factory itself is outside
    SURVEY section 8): affine (or additive) couplings on a float +-1 mask that flips from layer to layer, ResidualNet
    conditioners, no permutations.  Same construction order, so the seed reproduces the factory's weights
    (tests/golden/make_golden.py `realnvp` stores their checksums)."""
    from .transforms import AdditiveCouplingTransform
    from .nn.nets import ResidualNet
    if seed is not None:
        torch.manual_seed(seed)
    coupling = AdditiveCouplingTransform if use_volume_preserving else AffineCouplingTransform
    mask = torch.ones(features)
    mask[::2] = -1
    layers = []
    for _ in range(num_layers):
        layers.append(coupling(
            mask=mask,
            transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=hidden_features,
                                                               num_blocks=num_blocks_per_layer)))
        mask = mask * -1
    return Flow(CompositeTransform(layers), StandardNormal([features]))


def moons_maf_flow(num_layers=2, features=2, hidden_features=4, seed=0):
    """configs[0]: the reference README flow (README.md:41-51), MAF + RandomPermutation."""
    if seed is not None:
        torch.manual_seed(seed)
    layers = []
    for _ in range(num_layers):
        layers.append(MaskedAffineAutoregressiveTransform(features=features, hidden_features=hidden_features))
        layers.append(RandomPermutation(features=features))
    return Flow(CompositeTransform(layers), StandardNormal([features]))


def ar_rq_flow(features=784, hidden_features=256, num_bins=8, tail_bound=3.0, num_blocks=2, seed=0):
    """configs[4]: one MaskedPiecewiseRationalQuadraticAutoregressiveTransform."""
    if seed is not None:
        torch.manual_seed(seed)
    t = MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=features, hidden_features=hidden_features, num_bins=num_bins, tails="linear",
        tail_bound=tail_bound, num_blocks=num_blocks)
    return Flow(CompositeTransform([t]), StandardNormal([features]))


def conditional_rq_nsf_flow(num_layers=3, features=16, num_bins=8, hidden_features=128, raw_context=5,
                            context_features=12, tail_bound=3.0, seed=7, activation=torch.nn.functional.relu):
    """A conditional RQ-NSF coupling flow: ResidualNet conditioners with `context_features` (context
    concatenated in front of the initial layer, GLU gate per block: resnet.py:9-52, :92-100), the raw
    context embedded by a Linear (flows/base.py:42-49).  The construction order matches
    tests/golden/make_golden.py:conditional_flow_case, so the seed reproduces its weights."""
    if seed is not None:
        torch.manual_seed(seed)
    layers = []
    for i in range(num_layers):
        layers.append(RandomPermutation(features))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            mask=create_alternating_binary_mask(features, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: ResidualNet(
                i_, o_, hidden_features=hidden_features, context_features=context_features, num_blocks=2,
                activation=activation),
            num_bins=num_bins, tails="linear", tail_bound=tail_bound))
    return Flow(CompositeTransform(layers), StandardNormal([features]),
                embedding_net=torch.nn.Linear(raw_context, context_features))
