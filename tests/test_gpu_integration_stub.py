"""INTEGRATION.md section 2 as executable code: the ctypes stub a maintainer of the reference would add
(`tests/integration_stub/_amd.py`, listed verbatim in INTEGRATION.md) imported on its own -- it knows the C ABI
(`include/nflows_amd.h`) and torch, nothing of `nflows_amd` -- and run against the reference's vectors: the functional
seam (rational_quadratic.py:13-63) on every linear-tail case of rqs_functional.npz, the layer seam (coupling.py:82-98)
on the coupling fixtures."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from helpers import LAD_TOL, OUT_TOL, assert_fp32_parity, conditioning, parse_kwargs
from oracle import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "integration_stub", "_amd.py")


def test_integration_md_lists_the_stub_verbatim():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "```python\n" + open(STUB).read() + "```\n" in text


@pytest.fixture(scope="module")
def stub():
    os.environ.setdefault("NFLOWS_AMD_LIB", os.path.join(ROOT, "nflows_amd", "libnflows_amd.so"))
    spec = importlib.util.spec_from_file_location("_amd", STUB)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_functional_seam_against_the_reference_vectors(stub, golden_dir):
    g = np.load(os.path.join(golden_dir, "rqs_functional.npz"))
    ran = 0
    for name, inv, kw in g["meta"]:
        kw = parse_kwargs(kw)
        if kw.get("tails") != "linear":
            continue
        inverse = bool(int(inv))
        x, uw, uh, ud = (g[name + "/" + k] for k in ("x", "uw", "uh", "ud"))
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        y, lad = stub.unconstrained_rqs_hip(dev(x), dev(uw), dev(uh), dev(ud), inverse, kw["tail_bound"],
                                            kw.get("min_bin_width", 1e-3), kw.get("min_bin_height", 1e-3),
                                            kw.get("min_derivative", 1e-3), kw.get("enable_identity_init", False))
        y, lad = y.cpu().numpy(), lad.cpu().numpy()
        ospec = capi.make_spec(uw.shape[-1], **kw)
        cy, cl = conditioning(lambda *a: capi.rqs_elementwise(*a, ospec, inverse=inverse)[:2], (x, uw, uh, ud), (0, 1, 2, 3))
        assert_fp32_parity(y, g[name + "/y"], g[name + "/y64"], OUT_TOL, name + " y", cond=cy)
        assert_fp32_parity(lad, g[name + "/lad"], g[name + "/lad64"], LAD_TOL, name + " lad", cond=cl)
        tb = np.float32(kw["tail_bound"])
        outside = ~((x >= -tb) & (x <= tb))
        assert np.array_equal(y[outside].view(np.uint32), x[outside].view(np.uint32)) and np.all(lad[outside] == 0), name
        ran += 1
    assert ran >= 18
    assert int(stub._word(torch.device("cuda:0")).item()) & ~2 == 0     # (bit 2: a discriminant rounded below zero, as in the reference)


@pytest.mark.gpu
def test_layer_seam_against_the_reference_vectors(stub, golden_dir):
    g = np.load(os.path.join(golden_dir, "coupling.npz"))
    ran = 0
    for name, kind, cfg in g["meta"]:
        cfg = parse_kwargs(cfg)
        if kind != "rq" or cfg["tails"] != "linear":
            continue
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        x, params, tidx = g[name + "/x"], g[name + "/params"], g[name + "/transform_idx"]
        for direction, inverse in (("fwd", False), ("inv", True)):
            y, lad = stub.rqs_coupling_hip(dev(x), dev(params), dev(tidx), cfg["K"], cfg["tail_bound"], cfg["hidden"], inverse)
            ry, rl = g["%s/%s_y" % (name, direction)], g["%s/%s_lad" % (name, direction)]
            ry64, rl64 = g["%s/%s_y64" % (name, direction)], g["%s/%s_lad64" % (name, direction)]
            ospec = capi.make_spec(cfg["K"], tails="linear", tail_bound=cfg["tail_bound"], wh_divisor=float(np.sqrt(cfg["hidden"])))
            f64 = lambda xx, pp: capi.rqs_coupling(xx, pp, tidx, ospec, inverse=inverse)[:2]      # noqa: E731
            cy, cl = conditioning(f64, (x.astype(np.float64), params.astype(np.float64)), (0, 1))
            assert_fp32_parity(y.cpu().numpy(), ry, ry64, OUT_TOL, "%s %s y" % (name, direction), cond=cy)
            assert_fp32_parity(lad.cpu().numpy(), rl, rl64, LAD_TOL, "%s %s lad" % (name, direction), cond=cl)
            # untouched columns: bit-exact
            ident = np.setdiff1d(np.arange(cfg["D"]), tidx)
            assert np.array_equal(y.cpu().numpy()[:, ident].view(np.uint32), x[:, ident].view(np.uint32))
            ran += 1
    assert ran >= 8
