"""The reference's own unit tests for the path, restated against the drop-in classes on the device.

Every test below states the same scenario and the same assertion as one test of the reference's
suite (cited file:line, sizes and tolerances taken from there), written against `nflows_amd` with
every tensor on the HIP device.  The reference's test files themselves cannot be executed here: they
live in /root/reference, which does not exist on the GPU box, and its sources are not copied.

Covered (reference file -> section below):
  tests/transforms/transform_test.py          helpers `good`, `round_trip_is_identity`
  tests/transforms/coupling_test.py           affine / additive / piecewise couplings, vectors and images
  tests/transforms/splines/rational_quadratic_test.py
  tests/transforms/permutations_test.py
  tests/transforms/base_test.py               Composite / Multiscale / Inverse
  tests/flows/base_test.py
  tests/transforms/made_test.py
  tests/transforms/autoregressive_test.py
Left out, with the reason:
  * UMNNTransformTest (coupling_test.py:137-183) and MaskedUMNNAutoregressiveTranformTest
    (autoregressive_test.py:118-133): they need the third-party UMNN package (numerical integration
    of a network), which is not part of this path and not in the image (out of scope).
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BATCH = 10


# ---- tests/transforms/transform_test.py:10-31 ------------------------------------------------------
def good(tensor, shape=None):
    assert isinstance(tensor, torch.Tensor)
    assert not torch.isnan(tensor).any()
    assert not torch.isinf(tensor).any()
    if shape is not None:
        assert tensor.shape == torch.Size(shape)


def close(a, b, eps):
    """torchtestcase.assertEqual: same shape, max |a - b| <= eps (exact when eps is 0)."""
    assert a.shape == b.shape
    if eps:
        assert (a - b).abs().max().item() <= eps, (a - b).abs().max().item()
    else:
        assert torch.equal(a, b)


def different(a, b, eps=0.0):
    if eps:
        assert (a - b).abs().max().item() >= eps
    else:
        assert not torch.equal(a, b)


def round_trip_is_identity(transform, inputs, eps, logabsdet_eps=None):
    """transform_test.py:19-27: Composite([Inverse(t), t]) is the identity with zero log-determinant."""
    from nflows_amd.transforms import CompositeTransform, InverseTransform
    identity = CompositeTransform([InverseTransform(transform), transform])
    outputs, logabsdet = identity(inputs)
    good(outputs, inputs.shape)
    good(logabsdet, inputs.shape[:1])
    close(outputs, inputs, eps)
    close(logabsdet, torch.zeros(inputs.shape[:1], device=inputs.device), eps if logabsdet_eps is None else logabsdet_eps)


# ---- tests/transforms/coupling_test.py -------------------------------------------------------------
SHAPES = [[20], [2, 4, 4]]


def make_coupling(cls, shape, **kwargs):
    """coupling_test.py:13-33: mid-split mask, ResidualNet(30 wide, 5 blocks) for vectors,
    ConvResidualNet(16 channels) for images."""
    from nflows_amd.nn import nets
    from nflows_amd.utils import torchutils
    if len(shape) == 1:
        def create_net(n_in, n_out):
            return nets.ResidualNet(n_in, n_out, hidden_features=30, num_blocks=5)
    else:
        def create_net(n_in, n_out):
            return nets.ConvResidualNet(in_channels=n_in, out_channels=n_out, hidden_channels=16)
    mask = torchutils.create_mid_split_binary_mask(shape[0])
    return cls(mask=mask, transform_net_create_fn=create_net, **kwargs).to(DEV), mask.to(DEV)


def _coupling_classes():
    from nflows_amd.transforms import coupling
    return coupling


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", ["AffineCouplingTransform", "AdditiveCouplingTransform"])
def test_coupling_forward_keeps_identity_features(name, shape):
    """coupling_test.py:42-66 (affine), :92-118 (additive; its log-determinant is exactly zero)."""
    cls = getattr(_coupling_classes(), name)
    inputs = torch.randn(BATCH, *shape, device=DEV)
    transform, mask = make_coupling(cls, shape)
    outputs, logabsdet = transform(inputs)
    good(outputs, [BATCH] + shape)
    good(logabsdet, [BATCH])
    assert torch.equal(outputs[:, mask <= 0, ...], inputs[:, mask <= 0, ...])
    if name == "AdditiveCouplingTransform":
        assert torch.equal(logabsdet, torch.zeros(BATCH, device=DEV))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", ["AffineCouplingTransform", "AdditiveCouplingTransform"])
def test_coupling_round_trip(name, shape):
    """coupling_test.py:68-76, :120-128: eps 1e-6."""
    cls = getattr(_coupling_classes(), name)
    inputs = torch.randn(BATCH, *shape, device=DEV)
    transform, _ = make_coupling(cls, shape)
    round_trip_is_identity(transform, inputs, 1e-6)


@pytest.mark.parametrize("shape", SHAPES)
def test_affine_scale_activation_has_an_effect(shape):
    """coupling_test.py:78-89."""
    coupling = _coupling_classes()
    inputs = torch.randn(BATCH, *shape, device=DEV)
    transform, _ = make_coupling(coupling.AffineCouplingTransform, shape)
    out_default, lad_default = transform(inputs)
    transform.scale_activation = coupling.AffineCouplingTransform.GENERAL_SCALE_ACTIVATION
    out_general, lad_general = transform(inputs)
    different(out_default, out_general)
    different(lad_default, lad_general)


PIECEWISE = ["PiecewiseLinearCouplingTransform", "PiecewiseQuadraticCouplingTransform",
             "PiecewiseCubicCouplingTransform", "PiecewiseRationalQuadraticCouplingTransform"]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", PIECEWISE)
@pytest.mark.parametrize("unconstrained", [False, True])
def test_piecewise_coupling_forward(name, shape, unconstrained):
    """coupling_test.py:197-209 / :225-237 (inputs in [0, 1)), :211-223 / :239-251 (tails="linear",
    inputs 3 x normal)."""
    cls = getattr(_coupling_classes(), name)
    if unconstrained:
        inputs = 3.0 * torch.randn(BATCH, *shape, device=DEV)
        transform, mask = make_coupling(cls, shape, tails="linear")
    else:
        inputs = torch.rand(BATCH, *shape, device=DEV)
        transform, mask = make_coupling(cls, shape)
    outputs, logabsdet = transform(inputs)
    good(outputs, [BATCH] + shape)
    good(logabsdet, [BATCH])
    assert torch.equal(outputs[:, mask <= 0, ...], inputs[:, mask <= 0, ...])


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", PIECEWISE)
@pytest.mark.parametrize("unconstrained", [False, True])
def test_piecewise_coupling_round_trip(name, shape, unconstrained):
    """coupling_test.py:253-270: eps 1e-3."""
    cls = getattr(_coupling_classes(), name)
    if unconstrained:
        inputs = 3.0 * torch.randn(BATCH, *shape, device=DEV)
        transform, _ = make_coupling(cls, shape, tails="linear")
    else:
        inputs = torch.rand(BATCH, *shape, device=DEV)
        transform, _ = make_coupling(cls, shape)
    round_trip_is_identity(transform, inputs, 1e-3)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("name", PIECEWISE)
def test_piecewise_coupling_unconditional_transform_moves_identity_features(name, shape):
    """coupling_test.py:272-287."""
    cls = getattr(_coupling_classes(), name)
    inputs = torch.rand(BATCH, *shape, device=DEV)
    img_shape = shape[1:] if len(shape) > 1 else None
    transform, mask = make_coupling(cls, shape, apply_unconditional_transform=True, img_shape=img_shape)
    outputs, logabsdet = transform(inputs)
    good(outputs, [BATCH] + shape)
    good(logabsdet, [BATCH])
    different(outputs[:, mask <= 0, ...], inputs[:, mask <= 0, ...])


# ---- tests/transforms/splines/rational_quadratic_test.py ---------------------------------------------
def _spline_parameters(zeros=False, bins=10, shape=(2, 3, 4)):
    make = torch.zeros if zeros else torch.randn
    return (make(*shape, bins, device=DEV), make(*shape, bins, device=DEV), make(*shape, bins + 1, device=DEV))


def test_rational_quadratic_spline_round_trip():
    """rational_quadratic_test.py:8-31: inputs in [0, 1), eps 1e-3."""
    from nflows_amd.transforms import splines
    w, h, d = _spline_parameters()
    inputs = torch.rand(2, 3, 4, device=DEV)

    def call(x, inverse=False):
        return splines.rational_quadratic_spline(inputs=x, unnormalized_widths=w, unnormalized_heights=h,
                                                 unnormalized_derivatives=d, inverse=inverse)
    outputs, logabsdet = call(inputs)
    back, logabsdet_inv = call(outputs, inverse=True)
    close(inputs, back, 1e-3)
    close(logabsdet + logabsdet_inv, torch.zeros_like(logabsdet), 1e-3)


def test_rational_quadratic_spline_identity_init():
    """rational_quadratic_test.py:33-62: zero parameters + enable_identity_init = identity, eps 1e-6."""
    from nflows_amd.transforms import splines
    w, h, d = _spline_parameters(zeros=True)

    def call(x, inverse=False):
        return splines.rational_quadratic_spline(inputs=x, unnormalized_widths=w, unnormalized_heights=h,
                                                 unnormalized_derivatives=d, inverse=inverse,
                                                 enable_identity_init=True)
    for inverse in (False, True):
        inputs = torch.rand(2, 3, 4, device=DEV)
        outputs, logabsdet = call(inputs, inverse=inverse)
        close(inputs, outputs, 1e-6)
        close(logabsdet, torch.zeros_like(logabsdet), 1e-6)


@pytest.mark.parametrize("where", ["anywhere", "tails"])
def test_unconstrained_rational_quadratic_spline_round_trip(where):
    """rational_quadratic_test.py:65-88 (inputs 3 x normal), :90-114 (every input outside
    [-tail_bound, tail_bound], tail_bound = 1 = the function's default)."""
    from nflows_amd.transforms import splines
    w, h, d = _spline_parameters()
    shape = (2, 3, 4)
    if where == "anywhere":
        inputs = 3 * torch.randn(*shape, device=DEV)
    else:
        inputs = torch.sign(torch.randn(*shape, device=DEV)) * (1.0 + torch.rand(*shape, device=DEV))

    def call(x, inverse=False):
        return splines.unconstrained_rational_quadratic_spline(
            inputs=x, unnormalized_widths=w, unnormalized_heights=h, unnormalized_derivatives=d, inverse=inverse)
    outputs, logabsdet = call(inputs)
    back, logabsdet_inv = call(outputs, inverse=True)
    close(inputs, back, 1e-3)
    close(logabsdet + logabsdet_inv, torch.zeros_like(logabsdet), 1e-3)


def test_unconstrained_rational_quadratic_spline_identity_init():
    """rational_quadratic_test.py:116-146."""
    from nflows_amd.transforms import splines
    w, h, d = _spline_parameters(zeros=True)
    shape = (2, 3, 4)

    def call(x, inverse=False):
        return splines.unconstrained_rational_quadratic_spline(
            inputs=x, unnormalized_widths=w, unnormalized_heights=h, unnormalized_derivatives=d, inverse=inverse,
            enable_identity_init=True)
    inputs = torch.sign(torch.randn(*shape, device=DEV)) * (1.0 + torch.rand(*shape, device=DEV))
    outputs, logabsdet = call(inputs)
    close(inputs, outputs, 1e-6)
    close(logabsdet, torch.zeros_like(logabsdet), 1e-6)
    inputs = torch.rand(*shape, device=DEV)
    outputs, logabsdet = call(inputs, inverse=True)
    close(inputs, outputs, 1e-6)
    close(logabsdet, torch.zeros_like(logabsdet), 1e-6)


# ---- tests/transforms/permutations_test.py -----------------------------------------------------------
def test_permutation_forward_and_inverse():
    """permutations_test.py:12-36: exact equality (pure data movement)."""
    from nflows_amd.transforms import permutations
    features = 100
    inputs = torch.randn(BATCH, features, device=DEV)
    permutation = torch.randperm(features)
    transform = permutations.Permutation(permutation).to(DEV)
    outputs, logabsdet = transform(inputs)
    good(outputs, [BATCH, features])
    good(logabsdet, [BATCH])
    assert torch.equal(outputs, inputs[:, permutation.to(DEV)])
    assert torch.equal(logabsdet, torch.zeros(BATCH, device=DEV))
    back, logabsdet = transform.inverse(outputs)
    good(back, [BATCH, features])
    good(logabsdet, [BATCH])
    assert torch.equal(back, inputs)
    assert torch.equal(logabsdet, torch.zeros(BATCH, device=DEV))


def test_permutation_round_trips():
    """permutations_test.py:38-49."""
    from nflows_amd.transforms import permutations
    features = 100
    inputs = torch.randn(BATCH, features, device=DEV)
    for transform in (permutations.Permutation(torch.randperm(features)), permutations.RandomPermutation(features),
                      permutations.ReversePermutation(features)):
        round_trip_is_identity(transform.to(DEV), inputs, 0.0)


# ---- tests/transforms/base_test.py -------------------------------------------------------------------
def _scalar(scale):
    from nflows_amd.transforms import standard
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        return standard.AffineScalarTransform(scale=scale).to(DEV)


@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_composite_equals_the_product_of_its_scales(direction):
    """base_test.py:12-47: x2, identity, x0.25 = x0.5, both directions, exact."""
    from nflows_amd.transforms import CompositeTransform, standard
    shape = [2, 3, 4]
    inputs = torch.randn(BATCH, *shape, device=DEV)
    composite = CompositeTransform([_scalar(2.0), standard.IdentityTransform(), _scalar(0.25)])
    single = _scalar(0.5)
    run = (lambda t: t(inputs)) if direction == "forward" else (lambda t: t.inverse(inputs))
    outputs, logabsdet = run(composite)
    outputs_ref, logabsdet_ref = run(single)
    good(outputs, [BATCH] + shape)
    good(logabsdet, [BATCH])
    assert torch.equal(outputs, outputs_ref)
    close(logabsdet, logabsdet_ref, 1e-6)


def _multiscale(shape, split_dim=1):
    """base_test.py:51-61."""
    from nflows_amd.transforms import MultiscaleCompositeTransform
    mct = MultiscaleCompositeTransform(num_transforms=4, split_dim=split_dim)
    for scale in (2.0, 4.0, 0.5, 0.25):
        shape = mct.add_transform(_scalar(scale), shape)
    return mct


@pytest.mark.parametrize("shape", [(32, 4, 4), (64,), (65,)])
def test_multiscale_forward_flattens(shape):
    """base_test.py:63-71."""
    inputs = torch.ones(5, *shape, device=DEV)
    outputs, logabsdet = _multiscale(shape)(inputs)
    good(outputs, [5, int(np.prod(shape))])
    good(logabsdet, [5])


def test_multiscale_rejects_bad_shapes():
    """base_test.py:73-92: a dimension too small to halve three times, a split dimension the shape
    does not have, an inverse of inputs that are not flat."""
    with pytest.raises(ValueError):
        _multiscale((8,))
    with pytest.raises(ValueError):
        _multiscale([32], split_dim=2)
    with pytest.raises(ValueError):
        _multiscale([32, 4, 4]).inverse(torch.randn(5, 32, 4, 4, device=DEV))


@pytest.mark.parametrize("shape", [(32, 4, 4), (64,), (65,), (21,)])
def test_multiscale_round_trip(shape):
    """base_test.py:94-100."""
    transform = _multiscale(shape)
    inputs = torch.randn(5, *shape, device=DEV).view(5, -1)
    round_trip_is_identity(transform, inputs, 1e-6)


@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_inverse_transform_swaps_directions(direction):
    """base_test.py:103-128."""
    from nflows_amd.transforms import InverseTransform
    shape = [2, 3, 4]
    inputs = torch.randn(BATCH, *shape, device=DEV)
    transform = InverseTransform(_scalar(2.0))
    single = _scalar(0.5)
    run = (lambda t: t(inputs)) if direction == "forward" else (lambda t: t.inverse(inputs))
    outputs, logabsdet = run(transform)
    outputs_ref, logabsdet_ref = run(single)
    good(outputs, [BATCH] + shape)
    good(logabsdet, [BATCH])
    close(outputs, outputs_ref, 1e-6)
    close(logabsdet, logabsdet_ref, 1e-6)


# ---- tests/flows/base_test.py ------------------------------------------------------------------------
def _scalar_flow(shape):
    from nflows_amd.distributions.normal import StandardNormal
    from nflows_amd.flows import base
    return base.Flow(transform=_scalar(2.0), distribution=StandardNormal(shape)).to(DEV)


@pytest.mark.parametrize("with_context", [False, True])
def test_flow_log_prob_shape(with_context):
    """flows/base_test.py:13-29."""
    flow = _scalar_flow([2, 3, 4])
    inputs = torch.randn(BATCH, 2, 3, 4, device=DEV)
    context = torch.randn(BATCH, 5, 6, device=DEV) if with_context else None
    log_prob = flow.log_prob(inputs, context=context)
    assert isinstance(log_prob, torch.Tensor)
    assert log_prob.shape == torch.Size([BATCH])


@pytest.mark.parametrize("with_context", [False, True])
def test_flow_sample_shape(with_context):
    """flows/base_test.py:31-54."""
    flow = _scalar_flow([2, 3, 4])
    context = torch.randn(20, 5, 6, device=DEV) if with_context else None
    samples = flow.sample(10, context=context)
    assert isinstance(samples, torch.Tensor)
    assert samples.shape == torch.Size(([20] if with_context else []) + [10, 2, 3, 4])
    assert samples.device.type == "cuda"


def test_flow_sample_and_log_prob_agree_with_log_prob():
    """flows/base_test.py:56-73."""
    flow = _scalar_flow([2, 3, 4])
    samples, log_prob_1 = flow.sample_and_log_prob(10)
    log_prob_2 = flow.log_prob(samples)
    assert samples.shape == torch.Size([10, 2, 3, 4])
    assert log_prob_1.shape == torch.Size([10]) and log_prob_2.shape == torch.Size([10])
    close(log_prob_1, log_prob_2, 1e-4)


def test_flow_sample_and_log_prob_with_context():
    """flows/base_test.py:75-91."""
    flow = _scalar_flow([2, 3, 4])
    context = torch.randn(20, 5, 6, device=DEV)
    samples, log_prob = flow.sample_and_log_prob(10, context=context)
    assert samples.shape == torch.Size([20, 10, 2, 3, 4])
    assert log_prob.shape == torch.Size([20, 10])


@pytest.mark.parametrize("with_context", [False, True])
def test_flow_transform_to_noise_shape(with_context):
    """flows/base_test.py:93-108 (the context there has a different leading size than the inputs:
    this transform ignores it)."""
    flow = _scalar_flow([2, 3, 4])
    inputs = torch.randn(BATCH, 2, 3, 4, device=DEV)
    context = torch.randn(20, 5, 6, device=DEV) if with_context else None
    noise = flow.transform_to_noise(inputs, context=context)
    assert isinstance(noise, torch.Tensor)
    assert noise.shape == torch.Size([BATCH, 2, 3, 4])


# ---- tests/transforms/made_test.py -------------------------------------------------------------------
MADE_VARIANTS = [(False, False), (False, True), (True, False)]   # (use_residual_blocks, random_mask)


@pytest.mark.parametrize("use_residual_blocks,random_mask", MADE_VARIANTS)
@pytest.mark.parametrize("conditional", [True, False])
def test_made_output_shape(use_residual_blocks, random_mask, conditional):
    """made_test.py:12-44 (with 50 context features), :46-77 (without)."""
    from nflows_amd.transforms import made
    features, multiplier, batch = 100, 3, 16
    model = made.MADE(features=features, hidden_features=200, num_blocks=5, output_multiplier=multiplier,
                      context_features=50 if conditional else None, use_residual_blocks=use_residual_blocks,
                      random_mask=random_mask).to(DEV)
    inputs = torch.randn(batch, features, device=DEV)
    outputs = model(inputs, torch.randn(batch, 50, device=DEV)) if conditional else model(inputs)
    assert outputs.dim() == 2
    assert outputs.shape == (batch, multiplier * features)


@pytest.mark.parametrize("use_residual_blocks,random_mask", MADE_VARIANTS)
def test_made_outputs_depend_on_earlier_inputs_only(use_residual_blocks, random_mask):
    """made_test.py:81-107: the gradient of output k with respect to inputs k // multiplier and
    later is exactly zero (20 blocks, 256 wide)."""
    from nflows_amd.transforms import made
    features, multiplier = 10, 3
    model = made.MADE(features=features, hidden_features=256, num_blocks=20, output_multiplier=multiplier,
                      use_residual_blocks=use_residual_blocks, random_mask=random_mask).to(DEV)
    inputs = torch.randn(1, features, device=DEV, requires_grad=True)
    for k in range(features * multiplier):
        outputs = model(inputs)
        outputs[0, k].backward()
        depends = inputs.grad[0] != 0.0        # (gradients accumulate over k, as in the reference's loop)
        assert bool(torch.all(depends[k // multiplier:] == 0))


@pytest.mark.parametrize("use_residual_blocks", [True, False])
def test_made_sequential_masks_multiply_to_strictly_lower_triangular(use_residual_blocks):
    """made_test.py:109-137."""
    from nflows_amd.transforms import made
    features = 10
    model = made.MADE(features=features, hidden_features=50, num_blocks=5, output_multiplier=1,
                      use_residual_blocks=use_residual_blocks, random_mask=False).to(DEV)
    total = model.initial_layer.mask
    for block in model.blocks:
        if use_residual_blocks:
            assert isinstance(block, made.MaskedResidualBlock)
            total = block.linear_layers[0].mask @ total
            total = block.linear_layers[1].mask @ total
        else:
            assert isinstance(block, made.MaskedFeedforwardBlock)
            total = block.linear.mask @ total
    total = ((model.final_layer.mask @ total) > 0).float()
    assert torch.equal(total, torch.tril(torch.ones(features, features, device=DEV), -1))


def test_made_random_masks_stay_autoregressive():
    """made_test.py:139-160."""
    from nflows_amd.transforms import made
    features = 10
    model = made.MADE(features=features, hidden_features=50, num_blocks=5, output_multiplier=1,
                      use_residual_blocks=False, random_mask=True).to(DEV)
    total = model.initial_layer.mask
    for block in model.blocks:
        assert isinstance(block, made.MaskedFeedforwardBlock)
        total = block.linear.mask @ total
    total = ((model.final_layer.mask @ total) > 0).float()
    assert torch.equal(torch.triu(total), torch.zeros(features, features, device=DEV))


# ---- tests/transforms/autoregressive_test.py ---------------------------------------------------------
@pytest.mark.parametrize("use_residual_blocks,random_mask", MADE_VARIANTS)
def test_masked_affine_autoregressive(use_residual_blocks, random_mask):
    """autoregressive_test.py:12-78: forward and inverse give finite results of the right shape and
    invert one another (eps 1e-6)."""
    from nflows_amd.transforms import autoregressive
    features = 20
    inputs = torch.randn(BATCH, features, device=DEV)
    transform = autoregressive.MaskedAffineAutoregressiveTransform(
        features=features, hidden_features=30, num_blocks=5, use_residual_blocks=use_residual_blocks,
        random_mask=random_mask).to(DEV)
    for run in (transform, transform.inverse):
        outputs, logabsdet = run(inputs)
        good(outputs, [BATCH, features])
        good(logabsdet, [BATCH])
    # Outputs to the reference's 1e-6.  The log-determinant is a sum of 20 log-scales; the reference's
    # inverse loop ends on exactly the parameters its forward pass computes (same GEMM on the same
    # inputs), so its two sums cancel bit for bit -- here the inverse finds feature t from feature t's
    # own rows of the output layer (another summation order than the forward pass's full GEMM) and the
    # two sums differ by a few ulp of each term: measured 5e-7 .. 1.9e-6 (tools/maf_roundtrip_probe.py).
    round_trip_is_identity(transform, inputs, 1e-6, logabsdet_eps=4e-6)


@pytest.mark.parametrize("name,eps", [("MaskedPiecewiseLinearAutoregressiveTransform", 1e-3),
                                      ("MaskedPiecewiseQuadraticAutoregressiveTransform", 1e-4),
                                      ("MaskedPiecewiseCubicAutoregressiveTransform", 1e-3)])
def test_masked_piecewise_autoregressive_round_trip(name, eps):
    """autoregressive_test.py:81-98 (linear, 1e-3), :100-116 (quadratic, 1e-4), :136-152 (cubic, 1e-3):
    10 bins, inputs in [0, 1), residual blocks."""
    from nflows_amd.transforms import autoregressive
    features = 20
    inputs = torch.rand(BATCH, features, device=DEV)
    transform = getattr(autoregressive, name)(num_bins=10, features=features, hidden_features=30, num_blocks=5,
                                              use_residual_blocks=True).to(DEV)
    round_trip_is_identity(transform, inputs, eps)


def test_user_subclass_of_the_piecewise_coupling_base():
    """coupling.py:272-296: a `PiecewiseCouplingTransform` subclass that defines only `_piecewise_cdf` and
    `_transform_dim_multiplier` (the reference's extension point) runs the reference's sequence on the device -- split,
    conditioner, reshape per feature, the user's function, row-sum, merge --, forward and inverse, 2-D inputs."""
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import coupling as C
    from nflows_amd.utils import create_alternating_binary_mask

    class Shifted(C.PiecewiseCouplingTransform):
        def _transform_dim_multiplier(self):
            return 2

        def _piecewise_cdf(self, inputs, transform_params, inverse=False):
            a, b = transform_params[..., 0], transform_params[..., 1]
            if inverse:
                return (inputs - b) * torch.exp(-a), -a
            return inputs * torch.exp(a) + b, a

    torch.manual_seed(3)
    t = Shifted(create_alternating_binary_mask(10, even=True), lambda i, o: ResidualNet(i, o, 16, num_blocks=1)).to(DEV)
    x = torch.randn(64, 10, device=DEV)
    with torch.no_grad():
        y, lad = t(x)
        p = t.transform_net(x[:, t.identity_features]).reshape(64, 5, 2)
        assert torch.equal(y[:, t.identity_features], x[:, t.identity_features])
        assert torch.allclose(y[:, t.transform_features], x[:, t.transform_features] * torch.exp(p[..., 0]) + p[..., 1], atol=1e-6)
        assert torch.allclose(lad, p[..., 0].sum(1), atol=1e-6)
        xr, ladi = t.inverse(y)
        assert torch.allclose(xr, x, atol=1e-5) and torch.allclose(ladi, -lad, atol=1e-6)
