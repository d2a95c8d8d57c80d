"""Gradient parity of the HIP backward path against the reference's autograd
(tests/golden/grads.npz, produced by tests/golden/make_golden.py from the real reference) and
against autograd through the CPU eager port on fresh shapes.  Run with `-m gpu`.

Tolerance: like the forward tests, judged against the float64 gradient: the HIP fp32 gradient may
be at most 4x as far from it as the reference's own fp32 gradient, plus 2e-5 * scale."""
import os

import numpy as np
import pytest
import torch

from helpers import parse_kwargs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close_to_truth(got, ref32, ref64, what, tol=2e-5):
    got = got.detach().cpu().numpy().astype(np.float64)
    scale = 1.0 + np.abs(ref64).max()
    e_got = np.abs(got - ref64).max()
    e_ref = np.abs(ref32.astype(np.float64) - ref64).max()
    assert e_got <= 4 * e_ref + tol * scale, "%s: err %.3e (reference fp32 %.3e, scale %.2e)" % (what, e_got, e_ref, scale)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "grads.npz"))


def test_layer_gradients_match_reference_autograd(G):
    from nflows_amd import _native as N
    from nflows_amd import ops
    for name, kind, cfg in G["meta"]:
        if kind == "flow":
            continue
        cfg = parse_kwargs(cfg)
        x0, p0, tidx = G[name + "/x"], G[name + "/params"], dev(G[name + "/transform_idx"])
        Wy, Wl = dev(G[name + "/Wy"]), dev(G[name + "/Wl"])
        for direction, inv in (("fwd", False), ("inv", True)):
            x = dev(x0).requires_grad_(True)
            p = dev(p0).requires_grad_(True)
            if kind == "rq":
                H = cfg["hidden"]
                spec = ops.make_rqs_spec(cfg["K"], cfg["tails"], tail_bound=cfg["tail_bound"],
                                         wh_divisor=float(np.sqrt(H)) if H else 0.0)
                y, lad = ops.rqs_coupling(x, p, tidx, spec, inverse=inv)
            else:
                act = {"default": N.SCALE_DEFAULT, "general": N.SCALE_GENERAL, "additive": N.SCALE_ADDITIVE}[kind]
                y, lad = ops.affine_coupling(x, p, tidx, act, inverse=inv)
            ((y * Wy).sum() + (lad * Wl).sum()).backward()
            tag = "%s/%s" % (name, direction)
            close_to_truth(x.grad, G[tag + "_gx"], G[tag + "_gx64"], tag + " gx")
            close_to_truth(p.grad, G[tag + "_gp"], G[tag + "_gp64"], tag + " gparams")
    ops.check_status()


def test_flow_training_gradients_match_reference(G):
    from nflows_amd import configs
    name = "g_flow_nsf"
    cfg = parse_kwargs(dict((n, c) for n, _, c in G["meta"])[name])
    flow = configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"])
    prefix = name + "/sd/"
    flow.load_state_dict({k[len(prefix):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix)})
    flow = flow.to(DEV).train()
    for fuse in (True, False):
        flow._transform.fuse_permutations = fuse
        flow.zero_grad()
        x = dev(G[name + "/x"]).requires_grad_(True)
        loss = -flow.log_prob(x).mean()
        loss.backward()
        assert abs(loss.item() - float(G[name + "/loss64"])) <= 4 * abs(float(G[name + "/loss"]) - float(G[name + "/loss64"])) + 1e-5
        close_to_truth(x.grad, G[name + "/gx"], G[name + "/gx64"], "flow gx")
        for p_name, p in flow.named_parameters():
            close_to_truth(p.grad, G[name + "/grad/" + p_name], G[name + "/grad64/" + p_name], "flow " + p_name)


@pytest.mark.parametrize("B,D,K,tails,inverse,perm", [
    (257, 64, 8, "linear", False, "in"), (100, 64, 8, "linear", True, "out"), (33, 9, 5, "linear", False, None),
    (64, 12, 3, None, False, None), (64, 12, 10, None, True, None), (5, 784, 8, "linear", False, None),
    (3, 40, 16, "linear", True, "out"),
])
def test_rqs_coupling_gradients_vs_eager_autograd(B, D, K, tails, inverse, perm):
    """Fresh shapes (ragged batches, odd K, wide rows, fused permutations): HIP backward vs
    autograd through the CPU eager port in float64 (truth) and float32 (the reference's own path)."""
    from nflows_amd import ops
    from oracle import eager
    rng = np.random.RandomState(B * 7 + D + K)
    mask = rng.rand(D) < 0.5
    mask[0] = True
    tidx = np.nonzero(mask)[0]
    ident = np.nonzero(~mask)[0]
    dt = tidx.size
    P = 3 * K - 1 if tails == "linear" else 3 * K + 1
    x0 = (1.3 * rng.randn(B, D)).astype(np.float32) if tails == "linear" else rng.rand(B, D).astype(np.float32)
    p0 = rng.randn(B, dt * P).astype(np.float32)
    Wy, Wl = rng.randn(B, D).astype(np.float32), rng.randn(B).astype(np.float32)
    pm = rng.permutation(D) if perm else None
    H = 32

    def eager_loss(dtype):
        x = torch.from_numpy(x0).to(dtype).requires_grad_(True)
        p = torch.from_numpy(p0).to(dtype).requires_grad_(True)
        h = x[:, torch.from_numpy(pm)] if perm == "in" else x
        xt = h[:, tidx]
        pr = p.reshape(B, dt, P)
        uw, uh, ud = pr[..., :K] / np.sqrt(H), pr[..., K:2 * K] / np.sqrt(H), pr[..., 2 * K:]
        if tails == "linear":
            yt, l = eager.rqs_unconstrained(xt, uw, uh, ud, inverse=inverse, tail_bound=3.0)
        else:
            yt, l = eager.rqs_constrained(xt, uw, uh, ud, inverse=inverse)
        out = torch.empty_like(h)
        out[:, ident] = h[:, ident]
        out[:, tidx] = yt
        if perm == "out":
            out = out[:, torch.argsort(torch.from_numpy(pm))]
        loss = (out * torch.from_numpy(Wy).to(dtype)).sum() + (l.sum(1) * torch.from_numpy(Wl).to(dtype)).sum()
        loss.backward()
        return x.grad.numpy(), p.grad.numpy()

    gx32, gp32 = eager_loss(torch.float32)
    gx64, gp64 = eager_loss(torch.float64)
    x = dev(x0).requires_grad_(True)
    p = dev(p0).requires_grad_(True)
    spec = ops.make_rqs_spec(K, tails, tail_bound=3.0, wh_divisor=float(np.sqrt(H)))
    kw = {}
    if perm == "in":
        kw["in_perm"] = dev(pm.astype(np.int64))
    if perm == "out":
        kw["out_scatter"] = dev(pm.astype(np.int64))
    y, lad = ops.rqs_coupling(x, p, dev(tidx.astype(np.int64)), spec, inverse=inverse, **kw)
    ((y * dev(Wy)).sum() + (lad * dev(Wl)).sum()).backward()
    ops.check_status()
    close_to_truth(x.grad, gx32, gx64, "gx")
    close_to_truth(p.grad, gp32, gp64, "gparams")


def test_functional_and_autoregressive_gradients():
    from nflows_amd import configs
    from nflows_amd.transforms import splines
    from oracle import eager
    rng = np.random.RandomState(0)
    n, K = 300, 6
    x0, uw0, uh0, ud0 = (2 * rng.randn(n)).astype(np.float32), rng.randn(n, K).astype(np.float32), \
        rng.randn(n, K).astype(np.float32), rng.randn(n, K - 1).astype(np.float32)
    w = rng.randn(n).astype(np.float32)

    def run(fn, to, inv):
        ts = [to(a).requires_grad_(True) for a in (x0, uw0, uh0, ud0)]
        y, lad = fn(*ts, inverse=inv, tail_bound=2.0)
        ((y + 0.5 * lad) * to(w)).sum().backward()
        return [t.grad.detach().cpu().numpy() for t in ts]

    for inv in (False, True):
        got = run(lambda *a, **k: splines.unconstrained_rational_quadratic_spline(*a, tails="linear", **k), dev, inv)
        r32 = run(eager.rqs_unconstrained, lambda a: torch.from_numpy(a.copy()), inv)
        r64 = run(eager.rqs_unconstrained, lambda a: torch.from_numpy(a.copy()).double(), inv)
        for g_, a_, b_ in zip(got, r32, r64):
            close_to_truth(torch.from_numpy(g_), a_, b_, "functional grad")
    # MAF (cfg 1) and autoregressive RQ: one optimiser step runs and lowers the loss
    for flow in (configs.moons_maf_flow(), configs.ar_rq_flow(12, 32, 8, 3.0, 2)):
        flow = flow.to(DEV).train()
        opt = torch.optim.Adam(flow.parameters(), lr=1e-2)
        xb = torch.randn(256, flow._distribution._shape[0], device=DEV) * 0.5 + 0.3
        losses = []
        for _ in range(15):
            opt.zero_grad()
            loss = -flow.log_prob(xb).mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_training_loop_on_gpu_reduces_nll():
    """The reference's real call pattern (examples/moons.ipynb cell 3) on the drop-in flow."""
    from nflows_amd import configs
    torch.manual_seed(0)
    flow = configs.rq_nsf_flow(num_layers=4, features=8, num_bins=8, hidden_features=32).to(DEV).train()
    opt = torch.optim.Adam(flow.parameters(), lr=3e-3)
    data = torch.randn(4096, 8, device=DEV) * torch.linspace(0.3, 1.5, 8, device=DEV) + 0.7
    first = last = None
    for it in range(60):
        opt.zero_grad()
        loss = -flow.log_prob(data[torch.randint(0, 4096, (512,), device=DEV)]).mean()
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert last < first - 0.5
