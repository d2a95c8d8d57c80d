#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference (bayesiains/nflows at
/root/reference, imported read-only) on CPU.  Run in the build container only:

    python tests/golden/make_golden.py

The reference cannot travel to the GPU box, so the .npz files written next to this script are
committed; tests compare the C oracle (tests/test_oracle_golden.py) and the HIP path
(tests/test_gpu_golden.py) against them.  Nothing in nflows_amd imports this file.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SHIM = os.path.join(os.path.dirname(HERE), "_refshim")
sys.path.insert(0, SHIM)
sys.path.insert(0, REF)

import torch  # noqa: E402
from torch import nn  # noqa: E402

from nflows.transforms import splines  # noqa: E402
from nflows.transforms.base import CompositeTransform, InputOutsideDomain  # noqa: E402
from nflows.transforms.coupling import (  # noqa: E402
    AdditiveCouplingTransform,
    AffineCouplingTransform,
    PiecewiseCubicCouplingTransform,
    PiecewiseLinearCouplingTransform,
    PiecewiseQuadraticCouplingTransform,
    PiecewiseRationalQuadraticCouplingTransform,
)
from nflows.transforms.permutations import RandomPermutation, ReversePermutation  # noqa: E402
from nflows.transforms.autoregressive import (  # noqa: E402
    MaskedAffineAutoregressiveTransform,
    MaskedPiecewiseCubicAutoregressiveTransform,
    MaskedPiecewiseLinearAutoregressiveTransform,
    MaskedPiecewiseQuadraticAutoregressiveTransform,
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform,
)
from nflows.nn.nets import ResidualNet, ConvResidualNet, MLP  # noqa: E402
from nflows.flows.base import Flow  # noqa: E402
from nflows.distributions.normal import StandardNormal  # noqa: E402
from nflows.utils import torchutils  # noqa: E402

torch.set_num_threads(1)
warnings.filterwarnings("ignore")


def npy(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------- functional RQ
def rqs_cases():
    out = {}
    meta = []

    def run(name, x, uw, uh, ud, inverse, **kw):
        tails = kw.get("tails", "linear")
        if tails is None:
            fn = splines.rational_quadratic_spline
            kw = {k: v for k, v in kw.items() if k != "tails"}
        else:
            fn = splines.unconstrained_rational_quadratic_spline
        y, lad = fn(x.clone(), uw.clone(), uh.clone(), ud.clone(), inverse=inverse, **kw)
        y64, lad64 = fn(x.double(), uw.double(), uh.double(), ud.double(), inverse=inverse, **kw)
        out[name + "/x"] = npy(x)
        out[name + "/uw"] = npy(uw)
        out[name + "/uh"] = npy(uh)
        out[name + "/ud"] = npy(ud)
        out[name + "/y"] = npy(y)
        out[name + "/lad"] = npy(lad)
        out[name + "/y64"] = npy(y64)
        out[name + "/lad64"] = npy(lad64)
        kw2 = dict(kw)
        kw2["tails"] = tails
        meta.append((name, int(inverse), repr(sorted(kw2.items()))))

    g = torch.Generator().manual_seed(20260923)

    def edge(tb):
        f = np.float32
        vals = [-tb, tb, np.nextafter(f(tb), f(np.inf)), np.nextafter(f(-tb), f(-np.inf)),
                np.nextafter(f(tb), f(0)), np.nextafter(f(-tb), f(0)), float("nan"), 1e6, -1e6,
                0.0, -0.0, float("inf"), float("-inf")]
        return torch.tensor(vals, dtype=torch.float32)

    for tag, K, tb, n, scale in [("k8_tb3", 8, 3.0, 2048, 2.0), ("k10_tb1", 10, 1.0, 600, 1.0),
                                 ("k5_tb5", 5, 5.0, 512, 3.0), ("k2_tb3", 2, 3.0, 256, 1.0),
                                 ("k16_tb4", 16, 4.0, 300, 1.5)]:
        e = edge(tb)
        x = torch.cat([tb * torch.randn(n, generator=g), e])
        N = x.numel()
        uw = scale * torch.randn(N, K, generator=g)
        uh = scale * torch.randn(N, K, generator=g)
        ud = scale * torch.randn(N, K - 1, generator=g)
        for inv in (False, True):
            run("unc_%s_%s" % (tag, "inv" if inv else "fwd"), x, uw, uh, ud, inv, tails="linear",
                tail_bound=tb)

    # non-default minimums, identity-init beta
    K, tb = 8, 2.0
    x = tb * 0.8 * torch.randn(400, generator=g)
    uw, uh, ud = (torch.randn(400, K, generator=g), torch.randn(400, K, generator=g),
                  torch.randn(400, K - 1, generator=g))
    for inv in (False, True):
        run("unc_mins_%d" % inv, x, uw, uh, ud, inv, tails="linear", tail_bound=tb,
            min_bin_width=1e-2, min_bin_height=2e-2, min_derivative=5e-2)
        run("unc_idinit_%d" % inv, x, uw, uh, ud, inv, tails="linear", tail_bound=tb,
            enable_identity_init=True)
        run("unc_idinit_zero_%d" % inv, x, torch.zeros_like(uw), torch.zeros_like(uh),
            torch.zeros_like(ud), inv, tails="linear", tail_bound=tb, enable_identity_init=True)

    # extreme logits: softplus threshold (x*beta > 20) and saturated softmax
    K, tb = 8, 3.0
    x = tb * torch.randn(512, generator=g)
    uw = 12 * torch.randn(512, K, generator=g)
    uh = 12 * torch.randn(512, K, generator=g)
    ud = 15 * torch.randn(512, K - 1, generator=g)
    ud[:8, :] = torch.tensor([19.9, 20.0, 20.1, 25.0, -30.0, 40.0, 88.0, -88.0])[:, None]
    for inv in (False, True):
        run("unc_extreme_%d" % inv, x, uw, uh, ud, inv, tails="linear", tail_bound=tb)

    # everything in the tails (reference tests/transforms/splines/rational_quadratic_test.py:90-114)
    x = torch.cat([4 + torch.rand(64, generator=g), -4 - torch.rand(64, generator=g)])
    uw, uh, ud = (torch.randn(128, 8, generator=g), torch.randn(128, 8, generator=g),
                  torch.randn(128, 7, generator=g))
    for inv in (False, True):
        run("unc_alltails_%d" % inv, x, uw, uh, ud, inv, tails="linear", tail_bound=3.0)

    # constrained spline (tails=None): default unit box and a general box
    K = 10
    x = torch.rand(700, generator=g)
    x[:3] = torch.tensor([0.0, 1.0, 0.5])
    uw, uh, ud = (torch.randn(700, K, generator=g), torch.randn(700, K, generator=g),
                  torch.randn(700, K + 1, generator=g))
    for inv in (False, True):
        run("con_unit_%d" % inv, x, uw, uh, ud, inv, tails=None)
    K = 8
    # the reference checks inputs against [left, right] in both directions
    x = torch.rand(500, generator=g) * 2.0  # in [0, 2] = [left,right] & [bottom,top] overlap
    uw, uh, ud = (2 * torch.randn(500, K, generator=g), 2 * torch.randn(500, K, generator=g),
                  2 * torch.randn(500, K + 1, generator=g))
    for inv in (False, True):
        run("con_box_%d" % inv, x, uw, uh, ud, inv, tails=None, left=-1.0, right=2.0, bottom=0.0,
            top=5.0)

    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "rqs_functional.npz"), **out)
    print("rqs_functional:", len(meta), "cases")


# --------------------------------------------------------------------------- searchsorted
def searchsorted_case():
    # reference tests/utils/torchutils_test.py:80-90
    bin_locations = torch.linspace(0, 1, 10)
    left_boundaries = bin_locations[:-1]
    right_boundaries = bin_locations[:-1] + 0.1
    mid_points = bin_locations[:-1] + 0.05
    out = {"knots": npy(bin_locations)}
    for name, inp in [("left", left_boundaries), ("right", right_boundaries), ("mid", mid_points)]:
        idx = torchutils.searchsorted(bin_locations[None, :].clone(), inp)
        out[name + "_in"] = npy(inp)
        out[name + "_idx"] = npy(idx)
        assert torch.equal(idx, torch.arange(0, 9))
    np.savez_compressed(os.path.join(HERE, "searchsorted.npz"), **out)
    print("searchsorted ok")


# --------------------------------------------------------------------------- coupling layers
class FixedNet(nn.Module):
    """Stands in for the conditioner: returns a stored tensor (so the fixture pins the coupling
    arithmetic and not a GEMM)."""

    def __init__(self, table, hidden_features=None):
        super().__init__()
        self.table = table
        if hidden_features is not None:
            self.hidden_features = hidden_features

    def forward(self, identity, context=None):
        return self.table.to(identity.dtype).clone()


def coupling_cases():
    out = {}
    meta = []
    g = torch.Generator().manual_seed(77)

    def rq(name, D, mask, K, tails, tb, H, B, xgen):
        d_t = int((torch.as_tensor(mask) > 0).sum())
        P = 3 * K - 1 if tails == "linear" else 3 * K + 1
        table = 3.0 * torch.randn(B, d_t * P, generator=g)
        x = xgen(B, D)
        for dt_name, dt in (("", torch.float32), ("64", torch.float64)):
            for inv in (False, True):
                t = PiecewiseRationalQuadraticCouplingTransform(
                    mask, lambda i, o: FixedNet(table, H), num_bins=K, tails=tails, tail_bound=tb)
                fn = t.inverse if inv else t.forward
                y, lad = fn(x.to(dt))
                out["%s/%s_y%s" % (name, "inv" if inv else "fwd", dt_name)] = npy(y)
                out["%s/%s_lad%s" % (name, "inv" if inv else "fwd", dt_name)] = npy(lad)
        out[name + "/x"] = npy(x)
        out[name + "/params"] = npy(table)
        out[name + "/transform_idx"] = npy(t.transform_features)
        out[name + "/identity_idx"] = npy(t.identity_features)
        meta.append((name, "rq", repr(dict(D=D, K=K, tails=tails, tail_bound=tb, hidden=H, B=B))))

    rq("rq_d64_k8", 64, torchutils.create_alternating_binary_mask(64, even=True), 8, "linear", 3.0,
       128, 96, lambda B, D: torch.randn(B, D, generator=g) * 1.5)
    rq("rq_d64_k8_odd", 64, torchutils.create_alternating_binary_mask(64, even=False), 8, "linear",
       3.0, 128, 33, lambda B, D: torch.randn(B, D, generator=g) * 1.5)
    rq("rq_d7_k4_none", 7, torch.tensor([1, 0, 0, 1, 1, 0, 1]), 4, None, 1.0, None, 40,
       lambda B, D: torch.rand(B, D, generator=g))
    rq("rq_d10_k10_mid", 10, torchutils.create_mid_split_binary_mask(10), 10, "linear", 1.0, 30, 50,
       lambda B, D: torch.randn(B, D, generator=g))
    rq("rq_d5_k5", 5, torch.tensor([0, 1, 1, 0, 1]), 5, "linear", 2.0, 16, 257,
       lambda B, D: 2 * torch.randn(B, D, generator=g))
    rq("rq_d96_k6", 96, torchutils.create_alternating_binary_mask(96, even=True), 6, "linear", 4.0,
       64, 21, lambda B, D: 2 * torch.randn(B, D, generator=g))

    def affine(name, D, mask, B, kind):
        d_t = int((torch.as_tensor(mask) > 0).sum())
        mult = 1 if kind == "additive" else 2
        table = 2.0 * torch.randn(B, d_t * mult, generator=g)
        x = torch.randn(B, D, generator=g)
        for dt_name, dt in (("", torch.float32), ("64", torch.float64)):
            for inv in (False, True):
                if kind == "additive":
                    t = AdditiveCouplingTransform(mask, lambda i, o: FixedNet(table))
                elif kind == "general":
                    t = AffineCouplingTransform(
                        mask, lambda i, o: FixedNet(table),
                        scale_activation=AffineCouplingTransform.GENERAL_SCALE_ACTIVATION)
                else:
                    t = AffineCouplingTransform(mask, lambda i, o: FixedNet(table))
                fn = t.inverse if inv else t.forward
                y, lad = fn(x.to(dt))
                out["%s/%s_y%s" % (name, "inv" if inv else "fwd", dt_name)] = npy(y)
                out["%s/%s_lad%s" % (name, "inv" if inv else "fwd", dt_name)] = npy(lad)
        out[name + "/x"] = npy(x)
        out[name + "/params"] = npy(table)
        out[name + "/transform_idx"] = npy(t.transform_features)
        meta.append((name, "affine_" + kind, repr(dict(D=D, B=B))))

    for kind in ("default", "general", "additive"):
        affine("aff_d32_" + kind, 32, torchutils.create_alternating_binary_mask(32, even=True), 80, kind)
        affine("aff_d5_" + kind, 5, torch.tensor([1, 0, 1, 1, 0]), 37, kind)

    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "coupling.npz"), **out)
    print("coupling:", len(meta), "cases")


# --------------------------------------------------------------------------- whole flows
def state_to_np(prefix, module, out):
    for k, v in module.state_dict().items():
        out[prefix + "/sd/" + k] = npy(v)


class MLPConditioner(nn.Module):
    """coupling conditioners take (inputs, context); the reference MLP does not."""

    def __init__(self, i, o, hidden):
        super().__init__()
        self.mlp = MLP([i], [o], hidden)

    def forward(self, x, context=None):
        return self.mlp(x)


def flow_cases():
    out = {}
    meta = []

    def finish(name, flow, x, noise, cfg):
        flow.eval()
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, lad = flow._transform(x)
            xs, lad_inv = flow._transform.inverse(noise)
            f64 = flow.double()
            lp64 = f64.log_prob(x.double())
            z64, lad64 = f64._transform(x.double())
            xs64, ladi64 = f64._transform.inverse(noise.double())
            flow.float()
        state_to_np(name, flow, out)
        for k, v in dict(x=x, noise=noise, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                         log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64,
                         inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        meta.append((name, repr(cfg)))

    # cfg-4 shaped RQ-NSF coupling flow, shrunk
    for name, L, D, K, H, B in [("nsf_small", 4, 8, 4, 16, 64), ("nsf_d64", 2, 64, 8, 32, 40)]:
        torch.manual_seed(0)
        layers = []
        for i in range(L):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseRationalQuadraticCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=2),
                num_bins=K, tails="linear", tail_bound=3.0))
        flow = Flow(CompositeTransform(layers), StandardNormal([D]))
        # make the conditioner outputs non-trivial (default init gives near-identity splines)
        with torch.no_grad():
            for p_name, p in flow.named_parameters():
                if "final_layer" in p_name or "linear_layers.1" in p_name:
                    p.mul_(40.0) if "linear_layers.1" in p_name else p.mul_(6.0)
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(B, D, generator=g)
        noise = torch.randn(B, D, generator=g)
        finish(name, flow, x, noise, dict(kind="rq_nsf", L=L, D=D, K=K, H=H, B=B, tail_bound=3.0))

    # cfg-2 shaped affine coupling flow, shrunk
    torch.manual_seed(1)
    D, L, B = 12, 4, 50
    layers = []
    for i in range(L):
        layers.append(AffineCouplingTransform(
            mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: MLPConditioner(i_, o_, [24, 24])))
        layers.append(ReversePermutation(D))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]))
    g = torch.Generator().manual_seed(99)
    finish("affine_small", flow, torch.randn(B, D, generator=g), torch.randn(B, D, generator=g),
           dict(kind="affine", L=L, D=D, hidden=[24, 24], B=B))

    # cfg-1: README moons flow  (README.md:41-51 of the reference), 2 layers
    torch.manual_seed(2)
    layers = []
    for _ in range(2):
        layers.append(MaskedAffineAutoregressiveTransform(features=2, hidden_features=4))
        layers.append(RandomPermutation(features=2))
    flow = Flow(CompositeTransform(layers), StandardNormal([2]))
    g = torch.Generator().manual_seed(5)
    finish("moons_maf", flow, torch.randn(128, 2, generator=g), torch.randn(128, 2, generator=g),
           dict(kind="maf", L=2, D=2, H=4, B=128))

    # cfg-5 shaped autoregressive RQ spline, shrunk
    torch.manual_seed(3)
    D, K, H, B = 12, 8, 32, 48
    t = MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails="linear", tail_bound=3.0, num_blocks=2)
    with torch.no_grad():
        for p_name, p in t.named_parameters():
            if "final_layer" in p_name:
                p.mul_(5.0)
    flow = Flow(CompositeTransform([t]), StandardNormal([D]))
    g = torch.Generator().manual_seed(6)
    finish("ar_rq_small", flow, 1.5 * torch.randn(B, D, generator=g), torch.randn(B, D, generator=g),
           dict(kind="ar_rq", D=D, K=K, H=H, B=B, tail_bound=3.0, num_blocks=2))

    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows.npz"), **out)
    print("flows:", len(meta), "cases")


# --------------------------------------------------------------------------- permutations / misc
def misc_cases():
    out = {}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(33, 64, generator=g)
    perm = torch.randperm(64, generator=g)
    from nflows.transforms.permutations import Permutation
    p = Permutation(perm)
    y, lad = p(x)
    xi, _ = p.inverse(x)
    out.update(perm_x=npy(x), perm=npy(perm), perm_fwd=npy(y), perm_inv=npy(xi), perm_lad=npy(lad))
    v = torch.randn(70, 32, generator=g)
    out.update(rowsum_x=npy(v), rowsum=npy(torchutils.sum_except_batch(v)))
    n = StandardNormal([64])
    out.update(normal_lp=npy(n.log_prob(x)))
    for feats in (7, 8, 64):
        out["mask_alt_even_%d" % feats] = npy(torchutils.create_alternating_binary_mask(feats, even=True))
        out["mask_alt_odd_%d" % feats] = npy(torchutils.create_alternating_binary_mask(feats, even=False))
        out["mask_mid_%d" % feats] = npy(torchutils.create_mid_split_binary_mask(feats))
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
    print("misc ok")


def error_surface():
    """Record which exceptions the reference raises so tests can assert the same."""
    rec = {}
    x = torch.tensor([0.5, 1.5])
    uw = torch.zeros(2, 4)
    try:
        splines.rational_quadratic_spline(x, uw, uw, torch.zeros(2, 5))
    except InputOutsideDomain:
        rec["outside_domain"] = "InputOutsideDomain"
    try:
        splines.rational_quadratic_spline(torch.tensor([0.5]), torch.zeros(1, 4), torch.zeros(1, 4),
                                          torch.zeros(1, 5), min_bin_width=0.3)
    except ValueError as e:
        rec["min_width"] = str(e)
    try:
        splines.unconstrained_rational_quadratic_spline(torch.tensor([0.5]), torch.zeros(1, 4),
                                                        torch.zeros(1, 4), torch.zeros(1, 3),
                                                        tails="cubic")
    except RuntimeError as e:
        rec["tails"] = str(e)
    np.savez_compressed(os.path.join(HERE, "errors.npz"), **{k: np.array(v) for k, v in rec.items()})
    print("errors:", rec)


# --------------------------------------------------------------------------- shared-parameter CDF
def cdf_cases():
    from nflows.transforms.nonlinearities import PiecewiseRationalQuadraticCDF
    out = {}
    meta = []
    g = torch.Generator().manual_seed(31337)
    for name, F, K, tails, tb, B in [("cdf_f32_k8", 32, 8, "linear", 3.0, 70), ("cdf_f5_k10", 5, 10, None, 1.0, 33),
                                     ("cdf_f100_k4", 100, 4, "linear", 2.0, 19)]:
        torch.manual_seed(5)
        t = PiecewiseRationalQuadraticCDF([F], num_bins=K, tails=tails, tail_bound=tb)
        with torch.no_grad():
            for p in t.parameters():
                p.copy_(2.0 * torch.randn(p.shape, generator=g))
        x = (1.5 * torch.randn(B, F, generator=g)) if tails == "linear" else torch.rand(B, F, generator=g)
        for k, v in t.state_dict().items():
            out[name + "/sd/" + k] = npy(v)
        out[name + "/x"] = npy(x)
        for dt_name, dtp in (("", torch.float32), ("64", torch.float64)):
            tt = t.double() if dtp is torch.float64 else t.float()
            with torch.no_grad():
                for direction, fn in (("fwd", tt.forward), ("inv", tt.inverse)):
                    y, lad = fn(x.to(dtp))
                    out["%s/%s_y%s" % (name, direction, dt_name)] = npy(y)
                    out["%s/%s_lad%s" % (name, direction, dt_name)] = npy(lad)
        t.float()
        meta.append((name, repr(dict(F=F, K=K, tails=tails, tail_bound=tb, B=B))))
    # coupling layer with apply_unconditional_transform=True (coupling.py:524-535)
    torch.manual_seed(9)
    D, K, B = 16, 6, 40
    layer = PiecewiseRationalQuadraticCouplingTransform(
        torchutils.create_alternating_binary_mask(D), lambda i, o: ResidualNet(i, o, hidden_features=24),
        num_bins=K, tails="linear", tail_bound=3.0, apply_unconditional_transform=True)
    with torch.no_grad():
        for n_, p in layer.named_parameters():
            if "final_layer" in n_:
                p.mul_(5.0)
    x = torch.randn(B, D, generator=g)
    name = "coupling_uncond"
    for k, v in layer.state_dict().items():
        out[name + "/sd/" + k] = npy(v)
    out[name + "/x"] = npy(x)
    for dt_name, dtp in (("", torch.float32), ("64", torch.float64)):
        ll = layer.double() if dtp is torch.float64 else layer.float()
        with torch.no_grad():
            for direction, fn in (("fwd", ll.forward), ("inv", ll.inverse)):
                y, lad = fn(x.to(dtp))
                out["%s/%s_y%s" % (name, direction, dt_name)] = npy(y)
                out["%s/%s_lad%s" % (name, direction, dt_name)] = npy(lad)
    layer.float()
    meta.append((name, repr(dict(D=D, K=K, B=B, hidden=24))))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "cdf.npz"), **out)
    print("cdf:", len(meta), "cases")


# --------------------------------------------------------------------------- gradients
class ParamNet(nn.Module):
    """Conditioner stand-in whose output IS a parameter, so autograd yields d loss / d params."""

    def __init__(self, table, hidden_features=None):
        super().__init__()
        self.table = nn.Parameter(table.clone())
        if hidden_features is not None:
            self.hidden_features = hidden_features

    def forward(self, identity, context=None):
        return self.table * 1.0


def grad_cases():
    """Reference autograd through one coupling layer (loss = <y, Wy> + <logabsdet, Wl>) and through
    a whole flow (loss = -mean log_prob), in float32 and float64."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(4242)

    def layer(name, kind, D, mask, B, K=8, tails="linear", tb=3.0, H=128, scale=1.5):
        d_t = int((torch.as_tensor(mask) > 0).sum())
        if kind == "rq":
            P = 3 * K - 1 if tails == "linear" else 3 * K + 1
            table = scale * torch.randn(B, d_t * P, generator=g)
            x = (torch.randn(B, D, generator=g) * 1.4) if tails == "linear" else torch.rand(B, D, generator=g)
        else:
            table = torch.randn(B, d_t * (1 if kind == "additive" else 2), generator=g)
            x = torch.randn(B, D, generator=g)
        Wy = torch.randn(B, D, generator=g)
        Wl = torch.randn(B, generator=g)
        for dt_name, dtp in (("", torch.float32), ("64", torch.float64)):
            for inv in (False, True):
                net = ParamNet(table.to(dtp), H if kind == "rq" else None)
                if kind == "rq":
                    t = PiecewiseRationalQuadraticCouplingTransform(mask, lambda i, o: net, num_bins=K,
                                                                    tails=tails, tail_bound=tb)
                elif kind == "additive":
                    t = AdditiveCouplingTransform(mask, lambda i, o: net)
                elif kind == "general":
                    t = AffineCouplingTransform(mask, lambda i, o: net,
                                                scale_activation=AffineCouplingTransform.GENERAL_SCALE_ACTIVATION)
                else:
                    t = AffineCouplingTransform(mask, lambda i, o: net)
                xin = x.to(dtp).clone().requires_grad_(True)
                y, lad = (t.inverse if inv else t.forward)(xin)
                loss = (y * Wy.to(dtp)).sum() + (lad * Wl.to(dtp)).sum()
                loss.backward()
                tag = "%s/%s" % (name, "inv" if inv else "fwd")
                out[tag + "_gx" + dt_name] = npy(xin.grad)
                out[tag + "_gp" + dt_name] = npy(net.table.grad)
        out[name + "/x"] = npy(x)
        out[name + "/params"] = npy(table)
        out[name + "/Wy"] = npy(Wy)
        out[name + "/Wl"] = npy(Wl)
        out[name + "/transform_idx"] = npy(t.transform_features)
        meta.append((name, kind, repr(dict(D=D, K=K, tails=tails, tail_bound=tb, hidden=H, B=B))))

    layer("g_rq_d64_k8", "rq", 64, torchutils.create_alternating_binary_mask(64, even=True), 70)
    layer("g_rq_d10_k5", "rq", 10, torchutils.create_mid_split_binary_mask(10), 45, K=5, tb=2.0, H=16)
    layer("g_rq_d6_k4_none", "rq", 6, torch.tensor([1, 0, 1, 0, 0, 1]), 33, K=4, tails=None, tb=1.0, H=None,
          scale=1.0)
    layer("g_aff_default", "default", 12, torchutils.create_alternating_binary_mask(12, even=False), 40)
    layer("g_aff_general", "general", 12, torchutils.create_alternating_binary_mask(12, even=True), 40)
    layer("g_additive", "additive", 7, torch.tensor([1, 0, 1, 1, 0, 0, 1]), 21)

    # whole flow: -mean log_prob, gradients w.r.t. every parameter and the inputs.  Two instances: hidden width 16
    # (the layer-by-layer kernels), and -- drawn after it -- hidden width 128 on a batch of 128 rows (the shape the
    # conditioner's training kernels K14 take: this is their pin to the reference's own autograd)
    for name, seed, (L, D, K, H, B) in (("g_flow_nsf", 0, (3, 8, 4, 16, 48)), ("g_flow_nsf_h128", 1, (2, 16, 8, 128, 128))):
        flow_grad_case(out, meta, name, seed, L, D, K, H, B)
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "grads.npz"), **out)
    print("grads:", len(meta), "cases")


def flow_grad_case(out, meta, name, seed, L, D, K, H, B):
    if True:
        torch.manual_seed(seed)
        layers = []
        for i in range(L):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseRationalQuadraticCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=2),
                num_bins=K, tails="linear", tail_bound=3.0))
        flow = Flow(CompositeTransform(layers), StandardNormal([D]))
        with torch.no_grad():
            for p_name, p in flow.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(6.0)
                elif "linear_layers.1" in p_name:
                    p.mul_(40.0)
        xg = torch.randn(B, D, generator=torch.Generator().manual_seed(77 + seed))
        state_to_np(name, flow, out)
        out[name + "/x"] = npy(xg)
        for dt_name, dtp in (("", torch.float32), ("64", torch.float64)):
            f = flow.double() if dtp is torch.float64 else flow.float()
            f.zero_grad()
            xin = xg.to(dtp).clone().requires_grad_(True)
            loss = -f.log_prob(xin).mean()
            loss.backward()
            out[name + "/loss" + dt_name] = npy(loss)
            out[name + "/gx" + dt_name] = npy(xin.grad)
            for p_name, p in f.named_parameters():
                out[name + "/grad" + dt_name + "/" + p_name] = npy(p.grad)
        flow.float()
        meta.append((name, "flow", repr(dict(kind="rq_nsf", L=L, D=D, K=K, H=H, B=B, tail_bound=3.0))))


def sibling_spline_cases():
    """Linear and quadratic splines (splines/linear.py, splines/quadratic.py): constrained and
    unconstrained, forward and inverse, fp32 and fp64."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(2024)

    def finish(name, kind, fn, x, logits, kw):
        for dtp, suf in ((torch.float32, ""), (torch.float64, "64")):
            args = [t.to(dtp) for t in logits]
            for inverse in (False, True):
                y, lad = fn(x.to(dtp), *args, inverse=inverse, **kw)
                out["%s/%s%s" % (name, "inv_" if inverse else "", "y" + suf)] = npy(y)
                out["%s/%s%s" % (name, "inv_" if inverse else "", "lad" + suf)] = npy(lad)
        out[name + "/x"] = npy(x)
        for i, t in enumerate(logits):
            out["%s/logits%d" % (name, i)] = npy(t)
        meta.append((name, kind, repr(kw)))

    for K, n, scale in ((10, 400, 2.0), (4, 257, 4.0), (33, 300, 1.0)):
        # constrained: inputs in [0, 1] with the end points and bin boundaries included
        x = torch.rand(n, generator=g)
        x[:6] = torch.tensor([0.0, 1.0, 0.5, 1.0 / K, 1.0 - 1.0 / K, 1e-7])
        pdf = scale * torch.randn(n, K, generator=g)
        finish("lin_k%d" % K, "linear", splines.linear_spline, x, [pdf], {})
        uw = scale * torch.randn(n, K, generator=g)
        uh = scale * torch.randn(n, K + 1, generator=g)
        finish("quad_k%d" % K, "quadratic", splines.quadratic_spline, x, [uw, uh], {})
        # unconstrained: linear tails outside [-B, B]
        B = 3.0
        xu = 2.2 * torch.randn(n, generator=g)
        xu[:8] = torch.tensor([-B, B, 0.0, B + 1e-3, -B - 1e-3, float("nan"), 2.9999998, -2.9999998])
        finish("ulin_k%d" % K, "linear", splines.unconstrained_linear_spline, xu, [pdf],
               dict(tail_bound=B, tails="linear"))
        uh1 = scale * torch.randn(n, K - 1, generator=g)
        finish("uquad_k%d" % K, "quadratic", splines.unconstrained_quadratic_spline, xu, [uw, uh1],
               dict(tail_bound=B, tails="linear"))
    # shaped inputs and a non-default box
    x = 0.5 + torch.rand(3, 5, 7, generator=g)
    finish("lin_box", "linear", splines.linear_spline, x, [torch.randn(3, 5, 7, 6, generator=g)],
           dict(left=0.5, right=1.5, bottom=-1.0, top=2.0))
    x = 0.5 + torch.rand(3, 5, 7, generator=g)  # inverse of this case needs inputs in [bottom, top] = same box
    finish("quad_box", "quadratic", splines.quadratic_spline, x,
           [torch.randn(3, 5, 7, 6, generator=g), torch.randn(3, 5, 7, 7, generator=g)],
           dict(left=0.5, right=1.5, bottom=0.5, top=1.5, min_bin_width=1e-2, min_bin_height=1e-2))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "splines_lq.npz"), **out)
    print("sibling splines:", len(meta), "cases")


def sibling_spline_grad_cases():
    """Gradients of the linear / quadratic / cubic spline functionals by the reference's own autograd
    (fp32 and fp64): d(sum(y * Wy) + sum(lad * Wl)) / d(inputs, logits), forward and inverse."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(4242)

    def finish(name, kind, fn, x, logits, kw):
        wy = torch.randn(x.shape, generator=g)
        wl = torch.randn(x.shape, generator=g)
        for dtp, suf in ((torch.float32, ""), (torch.float64, "64")):
            for inverse in (False, True):
                xi = x.to(dtp).clone().requires_grad_(True)
                args = [t.to(dtp).clone().requires_grad_(True) for t in logits]
                y, lad = fn(xi, *args, inverse=inverse, **kw)
                ((y * wy.to(dtp)).sum() + (lad * wl.to(dtp)).sum()).backward()
                pre = "%s/%s" % (name, "inv_" if inverse else "")
                out[pre + "gx" + suf] = npy(xi.grad)
                for i, t in enumerate(args):
                    out["%sglogits%d%s" % (pre, i, suf)] = npy(t.grad)
        out[name + "/x"] = npy(x)
        out[name + "/wy"] = npy(wy)
        out[name + "/wl"] = npy(wl)
        for i, t in enumerate(logits):
            out["%s/logits%d" % (name, i)] = npy(t)
        meta.append((name, kind, repr(kw)))

    for K, n, scale in ((10, 300, 1.5), (4, 130, 2.5), (17, 200, 1.0)):
        x = 0.02 + 0.96 * torch.rand(n, generator=g)  # away from the clamp at the box ends
        pdf = scale * torch.randn(n, K, generator=g)
        finish("lin_k%d" % K, "linear", splines.linear_spline, x, [pdf], {})
        uw = scale * torch.randn(n, K, generator=g)
        uh = scale * torch.randn(n, K + 1, generator=g)
        finish("quad_k%d" % K, "quadratic", splines.quadratic_spline, x, [uw, uh], {})
        B = 3.0
        xu = 2.0 * torch.randn(n, generator=g)
        xu[:3] = torch.tensor([B + 0.5, -B - 0.25, 0.0])
        finish("ulin_k%d" % K, "linear", splines.unconstrained_linear_spline, xu, [pdf], dict(tail_bound=B, tails="linear"))
        uh1 = scale * torch.randn(n, K - 1, generator=g)
        finish("uquad_k%d" % K, "quadratic", splines.unconstrained_quadratic_spline, xu, [uw, uh1],
               dict(tail_bound=B, tails="linear"))
    for K, n, scale in ((10, 300, 1.0), (4, 130, 2.0), (8, 200, 0.5)):
        x = 0.02 + 0.96 * torch.rand(n, generator=g)
        logits = [scale * torch.randn(n, K, generator=g), scale * torch.randn(n, K, generator=g),
                  torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g)]
        finish("cub_k%d" % K, "cubic", splines.cubic_spline, x, logits, {})
        xu = 2.0 * torch.randn(n, generator=g)
        xu[:3] = torch.tensor([3.5, -3.25, 0.0])
        finish("ucub_k%d" % K, "cubic", splines.unconstrained_cubic_spline, xu, logits, dict(tail_bound=3.0, tails="linear"))
    x = 0.55 + 0.9 * torch.rand(3, 5, 7, generator=g)
    finish("lin_box", "linear", splines.linear_spline, x, [torch.randn(3, 5, 7, 6, generator=g)],
           dict(left=0.5, right=1.5, bottom=0.5, top=1.5))
    finish("quad_box", "quadratic", splines.quadratic_spline, x,
           [torch.randn(3, 5, 7, 6, generator=g), torch.randn(3, 5, 7, 7, generator=g)],
           dict(left=0.5, right=1.5, bottom=0.5, top=1.5, min_bin_width=1e-2, min_bin_height=1e-2))
    # K = 8 (the bin count of the benchmark configurations: the backward kernels have an instance compiled for
    # it); drawn after everything else so that the cases above keep their random numbers
    K, n, scale, B = 8, 260, 1.2, 3.0
    x = 0.02 + 0.96 * torch.rand(n, generator=g)
    pdf = scale * torch.randn(n, K, generator=g)
    uw = scale * torch.randn(n, K, generator=g)
    finish("lin_k8", "linear", splines.linear_spline, x, [pdf], {})
    finish("quad_k8", "quadratic", splines.quadratic_spline, x, [uw, scale * torch.randn(n, K + 1, generator=g)], {})
    xu = 2.0 * torch.randn(n, generator=g)
    xu[:3] = torch.tensor([B + 0.5, -B - 0.25, 0.0])
    finish("ulin_k8", "linear", splines.unconstrained_linear_spline, xu, [pdf], dict(tail_bound=B, tails="linear"))
    finish("uquad_k8", "quadratic", splines.unconstrained_quadratic_spline, xu,
           [uw, scale * torch.randn(n, K - 1, generator=g)], dict(tail_bound=B, tails="linear"))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "splines_lq_grads.npz"), **out)
    print("sibling spline gradients:", len(meta), "cases")


def sibling_coupling_cases():
    """Coupling layers outside the fused kernels: linear / quadratic piecewise couplings on [B, D]
    and spline couplings on [B, C, H, W] images with a ConvResidualNet conditioner."""
    out = {}
    meta = []

    def finish(name, t, x, noise, cfg):
        t.eval()
        with torch.no_grad():
            z, lad = t(x)
            xs, lad_inv = t.inverse(noise)
            t64 = t.double()
            z64, lad64 = t64(x.double())
            xs64, ladi64 = t64.inverse(noise.double())
            t.float()
        state_to_np(name, t, out)
        for k, v in dict(x=x, noise=noise, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv, z64=z64, lad64=lad64,
                         inv_x64=xs64, inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        meta.append((name, repr(cfg)))

    def sharpen(t, f_final, f_last):
        with torch.no_grad():
            for p_name, p in t.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(f_final)
                elif "linear_layers.1" in p_name or "conv_layers.1" in p_name:
                    p.mul_(f_last)

    g = torch.Generator().manual_seed(77)
    D, H, B, K = 6, 16, 50, 5
    for kind, cls, extra in (("linear", PiecewiseLinearCouplingTransform, {}),
                             ("quadratic", PiecewiseQuadraticCouplingTransform, {}),
                             ("quadratic_uncond", PiecewiseQuadraticCouplingTransform,
                              dict(apply_unconditional_transform=True))):
        torch.manual_seed(11)
        layers = []
        for i in range(2):
            layers.append(RandomPermutation(D))
            layers.append(cls(mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                              transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=1),
                              num_bins=K, tails="linear", tail_bound=3.0, **extra))
        t = CompositeTransform(layers)
        sharpen(t, 5.0, 30.0)
        finish("c2d_" + kind, t, 1.5 * torch.randn(B, D, generator=g), 1.5 * torch.randn(B, D, generator=g),
               dict(kind=kind, D=D, H=H, K=K, L=2, tail_bound=3.0, **extra))

    C, Hh, Ww, Bi, Kc = 4, 5, 3, 6, 4
    for kind, cls in (("rq", PiecewiseRationalQuadraticCouplingTransform), ("quadratic", PiecewiseQuadraticCouplingTransform),
                      ("linear", PiecewiseLinearCouplingTransform)):
        torch.manual_seed(12)
        t = cls(mask=torchutils.create_alternating_binary_mask(C, even=True),
                transform_net_create_fn=lambda i_, o_: ConvResidualNet(i_, o_, hidden_channels=8, num_blocks=1),
                num_bins=Kc, tails="linear", tail_bound=2.0)
        sharpen(t, 5.0, 30.0)
        finish("img_" + kind, t, torch.randn(Bi, C, Hh, Ww, generator=g), torch.randn(Bi, C, Hh, Ww, generator=g),
               dict(kind=kind, C=C, hidden_channels=8, K=Kc, tail_bound=2.0))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "couplings_lq.npz"), **out)
    print("sibling couplings:", len(meta), "cases")


def sibling_autoregressive_cases():
    """Masked autoregressive layers on the linear / quadratic / cubic splines
    (autoregressive.py:196-401), forward and the reference's D-pass inverse."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(515)
    D, H, B, K = 6, 16, 40, 5
    cases = [
        ("ar_linear", lambda: MaskedPiecewiseLinearAutoregressiveTransform(K, D, H), "unit"),
        ("ar_quadratic", lambda: MaskedPiecewiseQuadraticAutoregressiveTransform(D, H, num_bins=K, tails="linear",
                                                                                 tail_bound=3.0), "real"),
        ("ar_quadratic_box", lambda: MaskedPiecewiseQuadraticAutoregressiveTransform(D, H, num_bins=K), "unit"),
        ("ar_cubic", lambda: MaskedPiecewiseCubicAutoregressiveTransform(K, D, H), "unit"),
    ]
    for name, make, domain in cases:
        torch.manual_seed(31)
        t = make()
        with torch.no_grad():
            for p_name, p in t.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(3.0)
        if domain == "unit":
            x = 0.02 + 0.96 * torch.rand(B, D, generator=g)
            noise = 0.02 + 0.96 * torch.rand(B, D, generator=g)
        else:
            x = 1.5 * torch.randn(B, D, generator=g)
            noise = 1.5 * torch.randn(B, D, generator=g)
        t.eval()
        with torch.no_grad():
            z, lad = t(x)
            xs, lad_inv = t.inverse(noise)
            t64 = t.double()
            z64, lad64 = t64(x.double())
            xs64, ladi64 = t64.inverse(noise.double())
            t.float()
        state_to_np(name, t, out)
        for k, v in dict(x=x, noise=noise, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv, z64=z64, lad64=lad64,
                         inv_x64=xs64, inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        meta.append((name, repr(dict(D=D, H=H, K=K))))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "ar_siblings.npz"), **out)
    print("sibling autoregressive:", len(meta), "cases")


def cubic_coupling_cases():
    """PiecewiseCubicCouplingTransform on [B, D] (with and without the unconditional CDF on the
    identity half) and on images."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(909)

    def finish(name, t, x, noise, cfg):
        t.eval()
        with torch.no_grad():
            z, lad = t(x)
            xs, lad_inv = t.inverse(noise)
            t64 = t.double()
            z64, lad64 = t64(x.double())
            xs64, ladi64 = t64.inverse(noise.double())
            t.float()
        state_to_np(name, t, out)
        for k, v in dict(x=x, noise=noise, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv, z64=z64, lad64=lad64,
                         inv_x64=xs64, inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        meta.append((name, repr(cfg)))

    D, H, B, K = 6, 16, 50, 5
    for tag, extra in (("plain", {}), ("uncond", dict(apply_unconditional_transform=True))):
        torch.manual_seed(21)
        layers = []
        for i in range(2):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseCubicCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=1),
                num_bins=K, tails="linear", tail_bound=3.0, **extra))
        t = CompositeTransform(layers)
        with torch.no_grad():
            for p_name, p in t.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(3.0)
                elif "linear_layers.1" in p_name:
                    p.mul_(30.0)
        finish("c2d_cubic_" + tag, t, 1.5 * torch.randn(B, D, generator=g), 1.5 * torch.randn(B, D, generator=g),
               dict(kind="cubic", D=D, H=H, K=K, L=2, tail_bound=3.0, **extra))
    torch.manual_seed(22)
    C = 4
    t = PiecewiseCubicCouplingTransform(
        mask=torchutils.create_alternating_binary_mask(C, even=True),
        transform_net_create_fn=lambda i_, o_: ConvResidualNet(i_, o_, hidden_channels=8, num_blocks=1),
        num_bins=4, tails="linear", tail_bound=2.0)
    with torch.no_grad():
        for p_name, p in t.named_parameters():
            if "final_layer" in p_name:
                p.mul_(3.0)
            elif "conv_layers.1" in p_name:
                p.mul_(30.0)
    finish("img_cubic", t, torch.randn(6, C, 5, 3, generator=g), torch.randn(6, C, 5, 3, generator=g),
           dict(kind="cubic", C=C, hidden_channels=8, K=4, tail_bound=2.0))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "couplings_cubic.npz"), **out)
    print("cubic couplings:", len(meta), "cases")


def cubic_spline_cases():
    """Cubic spline (splines/cubic.py): constrained and linear tails, forward and inverse."""
    out = {}
    meta = []
    g = torch.Generator().manual_seed(31337)

    def finish(name, fn, x, logits, kw):
        for dtp, suf in ((torch.float32, ""), (torch.float64, "64")):
            args = [t.to(dtp) for t in logits]
            for inverse in (False, True):
                y, lad = fn(x.to(dtp), *args, inverse=inverse, **kw)
                out["%s/%s%s" % (name, "inv_" if inverse else "", "y" + suf)] = npy(y)
                out["%s/%s%s" % (name, "inv_" if inverse else "", "lad" + suf)] = npy(lad)
        out[name + "/x"] = npy(x)
        for i, t in enumerate(logits):
            out["%s/logits%d" % (name, i)] = npy(t)
        meta.append((name, "cubic", repr(kw)))

    for K, n, scale in ((10, 400, 1.0), (4, 257, 2.0), (8, 300, 0.5)):
        x = torch.rand(n, generator=g)
        x[:6] = torch.tensor([0.0, 1.0, 0.5, 1.0 / K, 1.0 - 1.0 / K, 1e-7])
        logits = [scale * torch.randn(n, K, generator=g), scale * torch.randn(n, K, generator=g),
                  torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g)]
        finish("cub_k%d" % K, splines.cubic_spline, x, logits, {})
        B = 3.0
        xu = 2.2 * torch.randn(n, generator=g)
        xu[:8] = torch.tensor([-B, B, 0.0, B + 1e-3, -B - 1e-3, float("nan"), 2.9999998, -2.9999998])
        finish("ucub_k%d" % K, splines.unconstrained_cubic_spline, xu, logits, dict(tail_bound=B, tails="linear"))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "splines_cubic.npz"), **out)
    print("cubic splines:", len(meta), "cases")


def flow_h128_case():
    """BASELINE layer shape (D = 64, K = 8, ResidualNet H = 128 x 2 blocks): the shape family the
    whole-layer kernel (K8) and the fused-final-Linear kernels (K7 / K7b) serve.  The weights are
    not stored (656 KB per layer): they are rebuilt from the seed, which the drop-in classes
    reproduce bit for bit (test_same_seed_same_weights_as_reference); per-parameter checksums are
    stored to make a drifted RNG visible."""
    out = {}
    L, D, K, H, B = 2, 64, 8, 128, 160
    torch.manual_seed(0)
    layers = []
    for i in range(L):
        layers.append(RandomPermutation(D))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=2),
            num_bins=K, tails="linear", tail_bound=3.0))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]))
    with torch.no_grad():
        for p_name, p in flow.named_parameters():
            if "final_layer" in p_name:
                p.mul_(4.0)
            elif "linear_layers.1" in p_name:
                p.mul_(30.0)
    g = torch.Generator().manual_seed(4321)
    x = 1.2 * torch.randn(B, D, generator=g)
    noise = torch.randn(B, D, generator=g)
    flow.eval()
    with torch.no_grad():
        lp = flow.log_prob(x)
        z, lad = flow._transform(x)
        xs, lad_inv = flow._transform.inverse(noise)
        f64 = flow.double()
        lp64 = f64.log_prob(x.double())
        z64, lad64 = f64._transform(x.double())
        xs64, ladi64 = f64._transform.inverse(noise.double())
        flow.float()
    name = "nsf_h128"
    for k, v in dict(x=x, noise=noise, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                     log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
        out[name + "/" + k] = npy(v)
    names, sums = [], []
    for k, v in flow.state_dict().items():
        names.append(k)
        sums.append([float(v.double().sum()), float(v.double().abs().sum())])
    out[name + "/param_names"] = np.array(names).astype(str)
    out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
    out["meta"] = np.array([(name, repr(dict(kind="rq_nsf", L=L, D=D, K=K, H=H, B=B, tail_bound=3.0,
                                              seed=0, scale_final=4.0, scale_linear1=30.0)))],
                           dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_h128.npz"), **out)
    print("flows_h128: 1 case")



def tails_none_flow_cases():
    """tails=None couplings (the constructor's default: the constrained spline on [0, 1]^2, K + 1 derivative logits,
    coupling.py:543-547, :565-570) with ResidualNet conditioners (H = 128 x 2 blocks) -- the shapes the whole-layer kernel
    K8 serves since round 6 (csrc/rqs_resnet_tails.hip).  Weights rebuilt from the seed (checksums stored)."""
    out, meta = {}, []
    for name, seed, (L, D, K, B) in (("none_k8_d16", 0, (3, 16, 8, 200)), ("none_k10_d64", 1, (2, 64, 10, 160)),
                                     ("none_k4_d24", 2, (4, 24, 4, 136))):
        torch.manual_seed(seed)
        layers = []
        for i in range(L):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseRationalQuadraticCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=128, num_blocks=2),
                num_bins=K, tails=None))
        t = CompositeTransform(layers)
        with torch.no_grad():
            for p_name, p in t.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(6.0)
        g = torch.Generator().manual_seed(100 + seed)
        x = 0.02 + 0.96 * torch.rand(B, D, generator=g)
        y = 0.02 + 0.96 * torch.rand(B, D, generator=g)
        t.eval()
        with torch.no_grad():
            z, lad = t(x)
            xi, ladi = t.inverse(y)
            t64 = t.double()
            z64, lad64 = t64(x.double())
            xi64, ladi64 = t64.inverse(y.double())
            t.float()
        for k, v in dict(x=x, y=y, z=z, lad=lad, inv_x=xi, inv_lad=ladi, z64=z64, lad64=lad64, inv_x64=xi64,
                         inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        names, sums = [], []
        for k, v in t.state_dict().items():
            names.append(k)
            sums.append([float(v.double().sum()), float(v.double().abs().sum())])
        out[name + "/param_names"] = np.array(names).astype(str)
        out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
        meta.append((name, repr(dict(L=L, D=D, K=K, H=128, B=B, seed=seed, scale_final=6.0))))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_tails_none.npz"), **out)
    print("flows_tails_none:", len(meta), "cases")


def conditional_flow_case():
    """A conditional flow at the whole-layer kernels' layer shape: RandomPermutation + RQ coupling with
    ResidualNet(H = 128, 2 blocks, context_features = 12) conditioners (resnet.py:9-52: context
    concatenated in front of the initial layer, GLU gate per block), context embedded by a Linear
    (flows/base.py:42-49).  Weights rebuilt from the seed (checksums stored), as in flow_h128_case."""
    out = {}
    L, D, K, H, B, C, E = 3, 16, 8, 128, 256, 5, 12
    torch.manual_seed(7)
    layers = []
    for i in range(L):
        layers.append(RandomPermutation(D))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, context_features=E,
                                                               num_blocks=2),
            num_bins=K, tails="linear", tail_bound=3.0))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]), embedding_net=nn.Linear(C, E))
    with torch.no_grad():
        for p_name, p in flow.named_parameters():
            if "final_layer" in p_name:
                p.mul_(4.0)
            elif "linear_layers.1" in p_name:
                p.mul_(30.0)
            elif "context_layer" in p_name:
                p.mul_(3.0)
    g = torch.Generator().manual_seed(77)
    x = 1.2 * torch.randn(B, D, generator=g)
    noise = torch.randn(B, D, generator=g)
    ctx = torch.randn(B, C, generator=g)
    flow.eval()
    with torch.no_grad():
        emb = flow._embedding_net(ctx)
        lp = flow.log_prob(x, context=ctx)
        z, lad = flow._transform(x, context=emb)
        xs, lad_inv = flow._transform.inverse(noise, context=emb)
        f64 = flow.double()
        emb64 = f64._embedding_net(ctx.double())
        lp64 = f64.log_prob(x.double(), context=ctx.double())
        z64, lad64 = f64._transform(x.double(), context=emb64)
        xs64, ladi64 = f64._transform.inverse(noise.double(), context=emb64)
        flow.float()
    name = "nsf_context_h128"
    for k, v in dict(x=x, noise=noise, context=ctx, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                     log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
        out[name + "/" + k] = npy(v)
    names, sums = [], []
    for k, v in flow.state_dict().items():
        names.append(k)
        sums.append([float(v.double().sum()), float(v.double().abs().sum())])
    out[name + "/param_names"] = np.array(names).astype(str)
    out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
    out["meta"] = np.array([(name, repr(dict(kind="rq_nsf_context", L=L, D=D, K=K, H=H, B=B, C=C, E=E, tail_bound=3.0,
                                              seed=7, scale_final=4.0, scale_linear1=30.0, scale_context=3.0)))],
                           dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_context.npz"), **out)
    print("flows_context: 1 case")


def conditional_flow_more_cases():
    """Conditional flows beyond 8 / 10 bins with ReLU (round 5: the whole-layer kernels' context instances for every served
    bin count and for the other block activations): the construction of conditional_flow_case with `num_bins` and the
    ResidualNets' `activation` varied, 256 rows, forward / inverse / log_prob in fp32 and fp64 from the REAL reference."""
    out, meta = {}, []
    cases = [("ctx_k4", 4, "relu"), ("ctx_k6", 6, "relu"), ("ctx_k9", 9, "relu"), ("ctx_k12", 12, "relu"), ("ctx_k16", 16, "relu"),
             ("ctx_k24", 24, "relu"), ("ctx_leaky_relu_k8", 8, "leaky_relu"), ("ctx_elu_k10", 10, "elu"), ("ctx_tanh_k8", 8, "tanh"),
             ("ctx_tanh_k10", 10, "tanh")]
    F = torch.nn.functional
    acts = {"relu": F.relu, "leaky_relu": F.leaky_relu, "elu": F.elu, "tanh": torch.tanh}
    for idx, (name, K, act) in enumerate(cases):
        L, D, H, B, C, E = 3, 16, 128, 256, 5, 12
        seed = 70 + idx
        torch.manual_seed(seed)
        layers = []
        for i in range(L):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseRationalQuadraticCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, context_features=E,
                                                                   num_blocks=2, activation=acts[act]),
                num_bins=K, tails="linear", tail_bound=3.0))
        flow = Flow(CompositeTransform(layers), StandardNormal([D]), embedding_net=nn.Linear(C, E))
        # (tanh / ELU blocks saturate: a milder sharpening than the ReLU cases' keeps the splines non-trivial)
        s_final, s_lin1, s_ctx = (4.0, 30.0, 3.0) if act in ("relu", "leaky_relu") else (8.0, 6.0, 3.0)
        with torch.no_grad():
            for p_name, p in flow.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(s_final)
                elif "linear_layers.1" in p_name:
                    p.mul_(s_lin1)
                elif "context_layer" in p_name:
                    p.mul_(s_ctx)
        g = torch.Generator().manual_seed(700 + idx)
        x = 1.2 * torch.randn(B, D, generator=g)
        noise = torch.randn(B, D, generator=g)
        ctx = torch.randn(B, C, generator=g)
        flow.eval()
        with torch.no_grad():
            emb = flow._embedding_net(ctx)
            lp = flow.log_prob(x, context=ctx)
            z, lad = flow._transform(x, context=emb)
            xs, lad_inv = flow._transform.inverse(noise, context=emb)
            f64 = flow.double()
            emb64 = f64._embedding_net(ctx.double())
            lp64 = f64.log_prob(x.double(), context=ctx.double())
            z64, lad64 = f64._transform(x.double(), context=emb64)
            xs64, ladi64 = f64._transform.inverse(noise.double(), context=emb64)
            flow.float()
        for k, v in dict(x=x, noise=noise, context=ctx, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                         log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        names, sums = [], []
        for k, v in flow.state_dict().items():
            names.append(k)
            sums.append([float(v.double().sum()), float(v.double().abs().sum())])
        out[name + "/param_names"] = np.array(names).astype(str)
        out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
        meta.append((name, repr(dict(kind="rq_nsf_context", L=L, D=D, K=K, H=H, B=B, C=C, E=E, tail_bound=3.0, seed=seed,
                                     activation=act, scale_final=s_final, scale_linear1=s_lin1, scale_context=s_ctx))))
        print("   %s: |lad| mean %.2f, reference fp32 vs fp64: z %.2e lad %.2e" % (
            name, float(lad64.abs().mean()), float((z.double() - z64).abs().max()), float((lad.double() - lad64).abs().max())))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_context_more.npz"), **out)
    print("flows_context_more: %d cases" % len(cases))


def realnvp_cases():
    """The reference's own RealNVP (flows/realnvp.py:17-71 `SimpleRealNVP`: affine / additive couplings on a flipping
    +-1 mask, ResidualNet conditioners, no permutations) at the whole-layer kernel K11's conditioner width, built by the
    FACTORY itself, weights sharpened (the blocks' second Linears are initialised to +-1e-3: x 100; final layers x 1.5) --
    forward / inverse / log_prob in fp32 and fp64, 256 rows.  Weights are rebuilt from the seed (checksums stored)."""
    from nflows.flows.realnvp import SimpleRealNVP
    out, meta = {}, []
    cases = [("realnvp_affine", dict(features=16, hidden_features=128, num_layers=6, num_blocks_per_layer=2), False),
             ("realnvp_additive", dict(features=16, hidden_features=128, num_layers=6, num_blocks_per_layer=2), True),
             ("realnvp_h64_d22", dict(features=22, hidden_features=64, num_layers=4, num_blocks_per_layer=1), False),
             ("realnvp_d64_b3", dict(features=64, hidden_features=128, num_layers=3, num_blocks_per_layer=3), False),
             # (40 identity features: the initial layer's four-k-step instances)
             ("realnvp_d80", dict(features=80, hidden_features=128, num_layers=3, num_blocks_per_layer=1), False)]
    for idx, (name, kw, vp) in enumerate(cases):
        seed = 90 + idx
        torch.manual_seed(seed)
        flow = SimpleRealNVP(use_volume_preserving=vp, **kw)
        with torch.no_grad():
            for p_name, p in flow.named_parameters():
                if "final_layer" in p_name:
                    p.mul_(1.5)
                elif "linear_layers.1" in p_name:
                    p.mul_(100.0)
        B, D = 256, kw["features"]
        g = torch.Generator().manual_seed(900 + idx)
        x = 1.2 * torch.randn(B, D, generator=g)
        noise = torch.randn(B, D, generator=g)
        flow.eval()
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, lad = flow._transform(x)
            xs, lad_inv = flow._transform.inverse(noise)
            f64 = flow.double()
            lp64 = f64.log_prob(x.double())
            z64, lad64 = f64._transform(x.double())
            xs64, ladi64 = f64._transform.inverse(noise.double())
            flow.float()
        for k, v in dict(x=x, noise=noise, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                         log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
            out[name + "/" + k] = npy(v)
        names, sums = [], []
        for k, v in flow.state_dict().items():
            names.append(k)
            sums.append([float(v.double().sum()), float(v.double().abs().sum())])
        out[name + "/param_names"] = np.array(names).astype(str)
        out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
        meta.append((name, repr(dict(kind="realnvp", B=B, seed=seed, use_volume_preserving=vp, scale_final=1.5,
                                     scale_linear1=100.0, **kw))))
        print("   %s: |lad| mean %.2f, |z - x| mean %.2f, reference fp32 vs fp64: z %.2e lad %.2e" % (
            name, float(lad64.abs().mean()), float((z64 - x.double()).abs().mean()),
            float((z.double() - z64).abs().max()), float((lad.double() - lad64).abs().max())))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_realnvp.npz"), **out)
    print("flows_realnvp: %d cases" % len(cases))


def bin_count_flow_cases():
    """Round 4, the whole-layer kernels' other bin counts (2 .. 16 except 8 and 10, and 20, 24, 32): two-layer coupling flows with steep
    splines (the recipe of steep_flow_cases) at D = 32, H = 128, forward and inverse of the reference in fp32 and fp64.
    tests/golden/flows_bins.npz; weights rebuilt from seed + steepen, checksums stored."""
    steep_flow_cases(file_name="flows_bins.npz", only_nsf=True, nsf_cases=tuple(
        ("bins_k%d" % K, 300 + K, 2, K, 60.0, 6.0, 10.0, 32, 128) for K in (2, 3, 4, 5, 6, 7, 9, 11, 12, 13, 16, 20, 24, 32)))


def trained_flow_case():
    """Round 4: a flow TRAINED with the reference (the recipe of examples/moons.ipynb cell 3: Adam on
    `-flow.log_prob(x).mean()`), not a seed-0 flow with scaled layers: six RandomPermutation + RQ coupling layers at
    D = 16, 8 bins, ResidualNet H = 64 x 2 blocks, 400 Adam steps (lr 3e-3, batch 512) on a 16-dimensional mixture of
    three correlated Gaussians pushed through a sinh-arcsinh warp -- multimodal, skewed, heavy-tailed: the conditioners
    learn steep and flat bins where the data asks for them.  Stored: the trained state_dict (700 KB), held-out x and
    noise, forward and inverse of the reference in fp32 and fp64, the loss before / after, the logit spreads per layer.
    tests/golden/flows_trained.npz."""
    sys.path.insert(0, os.path.dirname(HERE))
    L, D, K, H, B = 6, 16, 8, 64, 256
    torch.manual_seed(31)
    layers = []
    for i in range(L):
        layers.append(RandomPermutation(D))
        layers.append(PiecewiseRationalQuadraticCouplingTransform(
            mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=2),
            num_bins=K, tails="linear", tail_bound=3.0))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]))

    g = torch.Generator().manual_seed(131)
    means = 1.6 * torch.randn(3, D, generator=g)
    chol = [torch.tril(0.35 * torch.randn(D, D, generator=g)) + 0.5 * torch.eye(D) for _ in range(3)]

    def sample(n):
        comp = torch.randint(0, 3, (n,), generator=g)
        e = torch.randn(n, D, generator=g)
        xs = torch.stack([means[c] + chol[c] @ e[i] for i, c in enumerate(comp.tolist())])
        xs = torch.sinh(0.8 * torch.asinh(xs) + 0.25)     # skew + lighter / heavier tails per side
        return 0.45 * xs

    data = sample(8192)
    opt = torch.optim.Adam(flow.parameters(), lr=3e-3)
    flow.train()
    losses = []
    for step in range(400):
        idx = torch.randint(0, data.shape[0], (512,), generator=g)
        opt.zero_grad()
        loss = -flow.log_prob(data[idx]).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    flow.eval()
    x = sample(B)
    noise = torch.randn(B, D, generator=g)
    out = {}
    name = "trained_nsf"
    spread = []

    def hook(m, i, o):
        p = o.reshape(o.shape[0], -1, 3 * K - 1)
        spread.append((float((p[..., :2 * K] / np.sqrt(H)).std()), float(p[..., 2 * K:].std())))
    hooks = [t.transform_net.register_forward_hook(hook) for t in flow._transform._transforms if hasattr(t, "transform_net")]
    with torch.no_grad():
        flow._transform(x)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        lp = flow.log_prob(x)
        z, lad = flow._transform(x)
        xs, lad_inv = flow._transform.inverse(noise)
        f64 = flow.double()
        lp64 = f64.log_prob(x.double())
        z64, lad64 = f64._transform(x.double())
        xs64, ladi64 = f64._transform.inverse(noise.double())
        flow.float()
    for k, v in dict(x=x, noise=noise, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                     log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
        assert torch.isfinite(v).all(), k
        out[name + "/" + k] = npy(v)
    for k, v in flow.state_dict().items():
        out[name + "/sd/" + k] = npy(v)
    cfg = dict(kind="rq_nsf", L=L, D=D, K=K, H=H, B=B, tail_bound=3.0, seed=31, steps=400, lr=3e-3,
               loss_first=round(float(np.mean(losses[:5])), 3), loss_last=round(float(np.mean(losses[-20:])), 3),
               logit_std_wh_d_per_layer=[(round(a, 3), round(b, 3)) for a, b in spread])
    out["meta"] = np.array([(name, repr(cfg))], dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "flows_trained.npz"), **out)
    print("flows_trained:", cfg)
    print("reference fp32 error vs fp64: z max %.2e mean %.2e | inverse x max %.2e mean %.2e | held-out mean log_prob %.3f"
          % (float((z.double() - z64).abs().max()), float((z.double() - z64).abs().mean()),
             float((xs.double() - xs64).abs().max()), float((xs.double() - xs64).abs().mean()), float(lp64.mean())))


ACTIVATIONS = {"relu": torch.nn.functional.relu, "leaky_relu": torch.nn.functional.leaky_relu,
               "elu": torch.nn.functional.elu, "tanh": torch.tanh}


def activation_flow_cases():
    """Round 4: conditioners built with another activation than ReLU (resnet.py:27 `activation=`), steep two-layer
    coupling flows at D = 32, H = 128, 8 and 10 bins; tests/golden/flows_acts.npz."""
    steep_flow_cases(file_name="flows_acts.npz", only_nsf=True, nsf_cases=(
        ("act_leaky_relu_k8", 401, 2, 8, 60.0, 6.0, 10.0, 32, 128, "leaky_relu"),
        ("act_elu_k8", 402, 2, 8, 60.0, 6.0, 10.0, 32, 128, "elu"),
        ("act_tanh_k8", 403, 2, 8, 60.0, 6.0, 10.0, 32, 128, "tanh"),
        ("act_elu_k10", 404, 2, 10, 60.0, 6.0, 10.0, 32, 128, "elu"),
        ("act_tanh_k10", 405, 2, 10, 60.0, 6.0, 10.0, 32, 128, "tanh")))


STEEP_NSF_CASES = (("steep_nsf_k8", 21, 2, 8, 60.0, 6.0, 10.0, 64, 512), ("steep_nsf_k8_deep", 25, 4, 8, 20.0, 2.0, 10.0, 64, 512),
                   ("steep_nsf_k10", 22, 2, 10, 60.0, 6.0, 10.0, 64, 512))


def steep_flow_cases(file_name="flows_steep.npz", only_nsf=False, nsf_cases=STEEP_NSF_CASES):
    """Flows whose conditioner outputs are STEEP, as after training (round 4; every earlier whole-flow fixture is
    near-identity in the inverse direction, which hid a Newton step scaled by 1 / delta for two rounds):
    tests/helpers.py:steepen multiplies the width / height rows of every conditioner's output layer until the logits
    the spline sees are ~ N(0, 1 .. 2.5), the derivative rows likewise.  Forward (x -> z, logabsdet, log_prob) and INVERSE
    (noise -> x, logabsdet) of the reference in fp32 and fp64.  Weights are rebuilt from the seed + steepen() (the
    drop-in classes consume the RNG like the reference's); per-parameter checksums and the measured logit spreads
    are stored."""
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import steepen
    out = {}
    meta = []

    def logit_spread(flow, x, K, divisor):
        spread = []

        def hook(m, i, o):
            p = o.reshape(o.shape[0], -1, 3 * K - 1)
            spread.append((float((p[..., :2 * K] / divisor).std()), float(p[..., 2 * K:].std())))
        hooks = [t.register_forward_hook(hook) for t in nets(flow)]
        with torch.no_grad():
            flow._transform(x)
        for h in hooks:
            h.remove()
        return spread

    def nets(flow):
        for t in flow._transform._transforms:
            if hasattr(t, "transform_net"):
                yield t.transform_net
            elif hasattr(t, "autoregressive_net"):
                yield t.autoregressive_net

    def finish(name, flow, x, noise, cfg):
        flow.eval()
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, lad = flow._transform(x)
            xs, lad_inv = flow._transform.inverse(noise)
            f64 = flow.double()
            lp64 = f64.log_prob(x.double())
            z64, lad64 = f64._transform(x.double())
            xs64, ladi64 = f64._transform.inverse(noise.double())
            flow.float()
        for k, v in dict(x=x, noise=noise, log_prob=lp, z=z, lad=lad, inv_x=xs, inv_lad=lad_inv,
                         log_prob64=lp64, z64=z64, lad64=lad64, inv_x64=xs64, inv_lad64=ladi64).items():
            assert torch.isfinite(v).all(), (name, k)
            out[name + "/" + k] = npy(v)
        names, sums = [], []
        for k, v in flow.state_dict().items():
            names.append(k)
            sums.append([float(v.double().sum()), float(v.double().abs().sum())])
        out[name + "/param_names"] = np.array(names).astype(str)
        out[name + "/param_checksums"] = np.array(sums, dtype=np.float64)
        meta.append((name, repr(cfg)))
        print(name, "reference fp32 error vs fp64: z max %.2e mean %.2e | inverse x max %.2e mean %.2e"
              % (float((z.double() - z64).abs().max()), float((z.double() - z64).abs().mean()),
                 float((xs.double() - xs64).abs().max()), float((xs.double() - xs64).abs().mean())))

    # the BASELINE layer shape, 8 and 10 bins
    # (depth and steepness are chosen so that the flow stays INVERTIBLE in float64 -- inverse(forward(x)) = x to 1e-8 ..
    #  1e-13 --: six layers at logit spread 2 .. 3 compress volume by e^-500 and not even the float64 round trip
    #  returns x (measured: mean |error| 0.7), every fp32 error is then amplified chaotically and a defective
    #  refinement step drowns in the reference's own error.  Two layers at spread ~ 2 (every feature transformed once,
    #  sharp), four layers at spread ~ 1 (deep, round trip asserted), two layers of 10 bins.)
    for name, seed, L, K, wh, ds, hs, D, B, *rest in nsf_cases:
        H = 128
        act = rest[0] if rest else "relu"
        torch.manual_seed(seed)
        layers = []
        for i in range(L):
            layers.append(RandomPermutation(D))
            layers.append(PiecewiseRationalQuadraticCouplingTransform(
                mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
                transform_net_create_fn=lambda i_, o_: ResidualNet(i_, o_, hidden_features=H, num_blocks=2,
                                                                   activation=ACTIVATIONS[act]),
                num_bins=K, tails="linear", tail_bound=3.0))
        flow = Flow(CompositeTransform(layers), StandardNormal([D]))
        steepen(flow, K, wh, ds, hs)
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.randn(B, D, generator=g)
        noise = torch.randn(B, D, generator=g)
        spread = logit_spread(flow.eval(), x, K, float(np.sqrt(H)))
        finish(name, flow, x, noise, dict(kind="rq_nsf", L=L, D=D, K=K, H=H, B=B, tail_bound=3.0, seed=seed,
                                          wh_scale=wh, d_scale=ds, hidden_scale=hs, **({"activation": act} if rest else {}),
                                          logit_std_wh_d_per_layer=[(round(a, 3), round(b, 3)) for a, b in spread]))

    if only_nsf:
        out["meta"] = np.array(meta, dtype=object).astype(str)
        np.savez_compressed(os.path.join(HERE, file_name), **out)
        print(file_name, len(meta), "cases")
        return

    # affine analogue (configs[1]'s layer: AffineCouplingTransform + MLP [128, 128]): scale logits ~ N(0, 2)
    seed, L, D, B, ds = 23, 4, 32, 512, 8.0
    torch.manual_seed(seed)
    layers = []
    for i in range(L):
        layers.append(AffineCouplingTransform(
            mask=torchutils.create_alternating_binary_mask(D, even=(i % 2 == 0)),
            transform_net_create_fn=lambda i_, o_: MLPConditioner(i_, o_, [128, 128])))
    flow = Flow(CompositeTransform(layers), StandardNormal([D]))
    steepen(flow, None, d_scale=ds)
    g = torch.Generator().manual_seed(seed + 100)
    finish("steep_affine", flow, torch.randn(B, D, generator=g), torch.randn(B, D, generator=g),
           dict(kind="affine", L=L, D=D, hidden=[128, 128], B=B, seed=seed, d_scale=ds))

    # autoregressive RQ layer (configs[4]'s transform, shrunk): MADE has no hidden_features attribute, so the logits
    # are not divided (autoregressive.py:464-466)
    seed, D, K, H, B, wh, ds = 24, 40, 8, 64, 256, 10.0, 10.0
    torch.manual_seed(seed)
    t = MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails="linear", tail_bound=3.0, num_blocks=2)
    flow = Flow(CompositeTransform([t]), StandardNormal([D]))
    steepen(flow, K, wh, ds)
    g = torch.Generator().manual_seed(seed + 100)
    x = 1.5 * torch.randn(B, D, generator=g)
    noise = torch.randn(B, D, generator=g)
    spread = logit_spread(flow.eval(), x, K, 1.0)
    finish("steep_ar_rq", flow, x, noise,
           dict(kind="ar_rq", D=D, K=K, H=H, B=B, tail_bound=3.0, num_blocks=2, seed=seed, wh_scale=wh, d_scale=ds,
                logit_std_wh_d_per_layer=[(round(a, 3), round(b, 3)) for a, b in spread]))

    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, file_name), **out)
    print("flows_steep:", len(meta), "cases")


# --------------------------------------------------------------------------- the reference's bin index
class _SearchRecorder:
    """Wraps nflows.utils.torchutils.searchsorted for the duration of one spline call: the reference never returns
    `bin_idx` (rational_quadratic.py:115-118), so it is caught where it is made, together with the knots it was made
    from (before the in-place `+= eps` of torchutils.py:135)."""

    def __enter__(self):
        from nflows.transforms.splines import rational_quadratic as rq
        self.mod = rq.torchutils
        self.orig = self.mod.searchsorted
        self.calls = []

        def recording(bin_locations, inputs, eps=1e-6):
            knots = bin_locations.clone()
            idx = self.orig(bin_locations, inputs, eps)
            self.calls.append((knots, inputs.clone(), idx.clone()))
            return idx
        self.mod.searchsorted = recording
        return self

    def __exit__(self, *exc):
        self.mod.searchsorted = self.orig
        return False


def _reference_bins(x, uw, uh, ud, inverse, kw):
    """(y, lad, bin_idx int64 full shape with -1 where the reference does not search, knots of the searched axis
    [n_searched, K + 1], mask of the searched elements) of one call of the reference."""
    kw = dict(kw)
    tails = kw.pop("tails", "linear")
    if tails is None:
        fn = splines.rational_quadratic_spline
        searched = torch.ones_like(x, dtype=torch.bool)
    else:
        fn = splines.unconstrained_rational_quadratic_spline
        kw["tails"] = tails
        tb = kw.get("tail_bound", 1.0)
        searched = (x >= -tb) & (x <= tb)
    with _SearchRecorder() as rec:
        y, lad = fn(x.clone(), uw.clone(), uh.clone(), ud.clone(), inverse=inverse, **kw)
    full = torch.full(x.shape, -1, dtype=torch.int64)
    knots = torch.zeros(0, uw.shape[-1] + 1)
    if searched.any():
        assert len(rec.calls) == 1
        knots, xin, idx = rec.calls[0]
        assert torch.equal(xin, x[searched])
        full[searched] = idx
    else:
        assert len(rec.calls) == 0
    return y, lad, full, knots, searched


def bin_index_cases():
    """rqs_bins.npz: (1) the reference's bin_idx for every case of rqs_functional.npz (same inputs, read back from that
    file); (2) adversarial cases: inputs placed ON the reference's own knots of the searched axis and one ulp to either
    side (where an off-by-one in a fused search would show), 8 and 10 bins, both directions, mild and steep logits."""
    G = np.load(os.path.join(HERE, "rqs_functional.npz"))
    out = {}
    for name, inv, kw in G["meta"]:
        kwd = dict(eval(kw))
        x, uw, uh, ud = (torch.from_numpy(G[name + "/" + k]) for k in ("x", "uw", "uh", "ud"))
        y, lad, full, _, _ = _reference_bins(x, uw, uh, ud, bool(int(inv)), kwd)
        assert np.array_equal(npy(y).view(np.uint32), G[name + "/y"].view(np.uint32)), name   # the same call as the fixture's
        out[name + "/bin_idx"] = npy(full)
    meta = []
    g = torch.Generator().manual_seed(20260925)
    for tag, K, tb, n, scale in [("k8_tb3", 8, 3.0, 4096, 1.0), ("k8_tb3_steep", 8, 3.0, 4096, 3.0),
                                 ("k10_tb1", 10, 1.0, 2048, 2.0)]:
        uw = scale * torch.randn(n, K, generator=g)
        uh = scale * torch.randn(n, K, generator=g)
        ud = scale * torch.randn(n, K - 1, generator=g)
        for inv in (False, True):
            kw = dict(tails="linear", tail_bound=tb)
            x0 = (0.9 * tb * (2 * torch.rand(n, generator=g) - 1))
            _, _, _, knots, searched = _reference_bins(x0, uw, uh, ud, inv, kw)
            assert bool(searched.all())
            pick = torch.randint(0, K + 1, (n,), generator=g)
            on = knots[torch.arange(n), pick]
            f32 = np.float32
            below = torch.from_numpy(np.nextafter(npy(on), f32(-np.inf)))
            above = torch.from_numpy(np.nextafter(npy(on), f32(np.inf)))
            shift = torch.randint(0, 3, (n,), generator=g)
            x = torch.where(shift == 0, below, torch.where(shift == 1, on, above))
            if inv:
                # on a knot the reference's own discriminant can round below zero and its assertion
                # (rational_quadratic.py:142) then rejects the whole batch: such elements (found one by one) keep
                # their random position
                bad = []
                for i in range(n):
                    try:
                        _reference_bins(x[i:i + 1], uw[i:i + 1], uh[i:i + 1], ud[i:i + 1], inv, kw)
                    except AssertionError:
                        bad.append(i)
                if bad:
                    x[bad] = x0[bad]
                    shift[bad] = 3
                print("  %s inverse: %d of %d knot inputs trip the reference's discriminant assertion" % (tag, len(bad), n))
            y, lad, full, knots2, _ = _reference_bins(x, uw, uh, ud, inv, kw)
            name = "knots_%s_%s" % (tag, "inv" if inv else "fwd")
            y64, lad64 = splines.unconstrained_rational_quadratic_spline(x.double(), uw.double(), uh.double(), ud.double(),
                                                                         inverse=inv, **kw)
            for k_, v_ in (("x", x), ("uw", uw), ("uh", uh), ("ud", ud), ("y", y), ("lad", lad), ("y64", y64),
                           ("lad64", lad64), ("bin_idx", full), ("pick", pick), ("shift", shift)):
                out[name + "/" + k_] = npy(v_)
            # the searched elements' knots, scattered to the full shape (rows of elements in the tails: zeros)
            kn_full = torch.zeros(n, K + 1)
            inside = (x >= -tb) & (x <= tb)
            kn_full[inside] = knots2
            out[name + "/knots"] = npy(kn_full)
            meta.append((name, int(inv), repr(sorted(kw.items()))))
    out["meta"] = np.array(meta, dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "rqs_bins.npz"), **out)
    print("rqs_bins:", len(G["meta"]), "functional cases +", len(meta), "knot cases")


def config5_inverse_case():
    """BASELINE configs[4] by name: MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features=784,
    hidden_features=256, num_bins=8, tails="linear", tail_bound=3.0, num_blocks=2), the INVERSE (sampling) direction --
    the reference's 784-pass loop (autoregressive.py:43-52) on 64 rows in fp32 and fp64.  Until round 4 the GPU test ran
    that loop on the GPU box's CPU through the port (147 s of a 538 s suite); the real reference's result is a fixture
    now.  Weights from the seed (checksums stored), as configs.ar_rq_flow builds them."""
    torch.manual_seed(0)
    t = MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=784, hidden_features=256, num_bins=8, tails="linear", tail_bound=3.0, num_blocks=2)
    flow = Flow(CompositeTransform([t]), StandardNormal([784])).eval()
    z = torch.randn(4096, 784, generator=torch.Generator().manual_seed(4321))[:64]
    with torch.no_grad():
        x32, lad32 = flow._transform.inverse(z)
        f64 = flow.double()
        x64, lad64 = f64._transform.inverse(z.double())
        flow.float()
    out = {"cfg5/z": npy(z), "cfg5/inv_x": npy(x32), "cfg5/inv_lad": npy(lad32), "cfg5/inv_x64": npy(x64), "cfg5/inv_lad64": npy(lad64)}
    sums = [[float(v.double().sum()), float(v.double().abs().sum())] for v in flow.state_dict().values()]
    out["cfg5/param_checksums"] = np.array(sums, dtype=np.float64)
    out["meta"] = np.array([("cfg5", repr(dict(kind="ar_rq", D=784, H=256, K=8, tail_bound=3.0, num_blocks=2, seed=0, rows=64,
                                               batch_seed=4321)))], dtype=object).astype(str)
    np.savez_compressed(os.path.join(HERE, "config5_inverse.npz"), **out)
    print("config5_inverse: 64 rows")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
        config5_inverse_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "binidx":
        bin_index_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ar":
        sibling_autoregressive_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cubic":
        cubic_spline_cases()
        cubic_coupling_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lqgrads":
        sibling_spline_grad_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lq":
        sibling_spline_cases()
        sibling_coupling_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tails_none":
        tails_none_flow_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "h128":
        flow_h128_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "context":
        conditional_flow_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "realnvp":
        realnvp_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "context_more":
        conditional_flow_more_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "grads":
        grad_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "steep":
        steep_flow_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "trained":
        trained_flow_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "acts":
        activation_flow_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bins":
        bin_count_flow_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cdf":
        cdf_cases()
        sys.exit(0)
    rqs_cases()
    searchsorted_case()
    coupling_cases()
    flow_cases()
    misc_cases()
    error_surface()
    grad_cases()
    cdf_cases()
    flow_h128_case()
    sibling_spline_cases()
    sibling_spline_grad_cases()
    sibling_coupling_cases()
    cubic_spline_cases()
    cubic_coupling_cases()
    sibling_autoregressive_cases()
    steep_flow_cases()
    bin_count_flow_cases()
    activation_flow_cases()
    trained_flow_case()
    bin_index_cases()
    config5_inverse_case()
