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


def test_float64_functional_gradients_match_reference_autograd(G):
    """K5d-backward (nfa_rqs_elementwise_backward_f64) against the reference's float64 autograd through a spline
    coupling layer (grads.npz, loss = <y, Wy> + <logabsdet, Wl>): (a) the functional on strided views of the
    packed table, (b) the whole layer class in double (torch gathers + the functional + index copies), both
    directions; linear tails with the conditioner's divisor (K = 8, 5) and the constrained spline (K = 4)."""
    from nflows_amd import ops
    from nflows_amd import transforms as T

    class Table(torch.nn.Module):
        def __init__(self, table, hidden):
            super().__init__()
            self.table = torch.nn.Parameter(table)
            if hidden is not None:
                self.hidden_features = hidden

        def forward(self, inputs, context=None):
            return self.table

    for name, kind, cfg in G["meta"]:
        if kind != "rq":
            continue
        cfg = parse_kwargs(cfg)
        K, tails, tb, H = cfg["K"], cfg["tails"], cfg["tail_bound"], cfg["hidden"]
        P = 3 * K - 1 if tails == "linear" else 3 * K + 1
        x0, p0 = G[name + "/x"].astype(np.float64), G[name + "/params"].astype(np.float64)
        tidx_np = G[name + "/transform_idx"]
        tidx = dev(tidx_np)
        Wy, Wl = dev(G[name + "/Wy"].astype(np.float64)), dev(G[name + "/Wl"].astype(np.float64))
        B, D = x0.shape
        dt = len(tidx_np)
        mask = np.zeros(D, dtype=np.int64)
        mask[tidx_np] = 1
        spec = ops.make_rqs_spec(K, tails, tail_bound=tb, wh_divisor=float(np.sqrt(H)) if H else 0.0)
        for direction, inv in (("fwd", False), ("inv", True)):
            tag = "%s/%s" % (name, direction)
            ref_gx, ref_gp = G[tag + "_gx64"], G[tag + "_gp64"]
            # (a) the functional
            xt = dev(x0[:, tidx_np]).requires_grad_(True)
            p = dev(p0).requires_grad_(True)
            pr = p.view(B, dt, P)
            y, lad = ops.rqs_elementwise(xt, pr[..., :K], pr[..., K:2 * K], pr[..., 2 * K:], spec, inverse=inv)
            assert y.dtype == torch.float64
            ((y * Wy[:, tidx]).sum() + (lad.sum(dim=1) * Wl).sum()).backward()
            for got, ref, what in ((xt.grad, ref_gx[:, tidx_np], "gx"), (p.grad, ref_gp, "gparams")):
                err = (got.cpu().numpy() - ref).__abs__().max()
                assert err <= 1e-10 * (1 + np.abs(ref).max()), "%s functional %s: %.3e" % (tag, what, err)
            # (b) the layer class in double
            net = Table(dev(p0), H)
            layer = T.PiecewiseRationalQuadraticCouplingTransform(torch.from_numpy(mask), lambda i, o: net, num_bins=K,
                                                                  tails=tails, tail_bound=tb).to(DEV).double()
            x = dev(x0).requires_grad_(True)
            y, lad = (layer.inverse if inv else layer.forward)(x)
            ((y * Wy).sum() + (lad * Wl).sum()).backward()
            for got, ref, what in ((x.grad, ref_gx, "gx"), (net.table.grad, ref_gp, "gparams")):
                err = (got.cpu().numpy() - ref).__abs__().max()
                assert err <= 1e-10 * (1 + np.abs(ref).max()), "%s layer %s: %.3e" % (tag, what, err)
    ops.check_status()


def test_float64_flow_trains_on_the_device():
    """`flow.double()` takes optimisation steps on the GPU (the reference is dtype-generic; before K5d-backward the
    float64 functional refused to differentiate) and torch.autograd.gradcheck accepts the spline layer."""
    from nflows_amd import transforms as T
    from nflows_amd.distributions import StandardNormal
    from nflows_amd.flows import Flow
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.utils import torchutils
    from nflows_amd.transforms import splines
    torch.manual_seed(3)
    layers = []
    for i in range(2):
        layers.append(T.ReversePermutation(4))
        layers.append(T.PiecewiseRationalQuadraticCouplingTransform(
            torchutils.create_alternating_binary_mask(4, even=(i % 2 == 0)),
            lambda a, b: ResidualNet(a, b, 16, num_blocks=1), num_bins=6, tails="linear", tail_bound=3.0))
    flow = Flow(T.CompositeTransform(layers), StandardNormal([4])).to(DEV).double().train()
    opt = torch.optim.Adam(flow.parameters(), lr=1e-2)
    data = torch.randn(512, 4, device=DEV, dtype=torch.float64) * 0.5 + 0.3
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = -flow.log_prob(data).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, losses[::10]
    g = torch.Generator().manual_seed(9)
    x = (2.0 * torch.randn(12, generator=g, dtype=torch.float64)).to(DEV).requires_grad_(True)
    uw, uh = (torch.randn(12, 5, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True) for _ in range(2))
    ud = torch.randn(12, 4, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True)
    for inverse in (False, True):
        assert torch.autograd.gradcheck(
            lambda *a: splines.unconstrained_rational_quadratic_spline(*a, inverse=inverse, tails="linear", tail_bound=2.5),
            (x, uw, uh, ud), eps=1e-6, atol=1e-6, rtol=1e-5)


def _k14_net(B, di, nb, H=128):
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(B + di + nb)
    net = ResidualNet(di, 40, H, num_blocks=nb).to(DEV)
    with torch.no_grad():   # (blocks end in U(-1e-3, 1e-3) layers: scale them up so that every path carries signal)
        for b in net.blocks:
            b.linear_layers[1].weight.mul_(60.0)
            b.linear_layers[1].bias.mul_(60.0)
    return net


@pytest.mark.parametrize("B,di,nb,H", [(256, 32, 2, 128), (384, 12, 1, 128), (128, 64, 3, 128), (1024, 36, 0, 128),
                                       (256, 8, 2, 64), (128, 20, 1, 52), (256, 3, 2, 128), (128, 21, 1, 64), (128, 63, 2, 128)])
def test_fused_conditioner_training_kernels(B, di, nb, H, monkeypatch):
    """K14 (nfa_resnet_hidden_forward_f32 / _backward_f32 + K10) against autograd through the eager modules with the
    same weights: the net's output, the input gradient and every parameter gradient, judged against float64 -- at
    most 4 x the eager fp32 path's own error + 1e-6 of the scale.  One / two / four k-steps of identity features,
    zero to three blocks, hidden widths below 128 (zero-padded into the kernels' 128), identity-feature counts that are
    not multiples of four (3, 21, 63: zero columns on the host).  (Small batches: an activation within rounding of zero flips a ReLU mask and moves a
    gradient by O(weight) in ANY fp32 implementation; the large-batch test below pins the masks instead.)"""
    import copy
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    net = _k14_net(B, di, nb, H)
    x = torch.randn(B, di, device=DEV, requires_grad=True)
    w = torch.randn(B, 40, device=DEV)
    calls = []
    real = ops.resnet_hidden_forward
    monkeypatch.setattr(ops, "resnet_hidden_forward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(ResidualNet, "fuse_training", fused)
        net.zero_grad(set_to_none=True)
        x.grad = None
        out = net(x)
        (out * w).sum().backward()
        return [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()]

    got = run(True)
    assert len(calls) == 1
    eager = run(False)
    assert len(calls) == 1
    net64 = copy.deepcopy(net).double()
    x64 = x.detach().double().requires_grad_(True)
    out64 = net64(x64)
    (out64 * w.double()).sum().backward()
    truth = [out64.detach(), x64.grad] + [p.grad for p in net64.parameters()]
    names = ["output", "grad_inputs"] + [n for n, _ in net.named_parameters()]
    for name, a, b, t in zip(names, got, eager, truth):
        assert a.shape == t.shape, name
        scale = 1.0 + t.abs().max().item()
        e_a, e_b = (a.double() - t).abs().max().item(), (b.double() - t).abs().max().item()
        assert e_a <= 4 * e_b + 1e-6 * scale, "%s: fused %.3e, eager %.3e, scale %.2e" % (name, e_a, e_b, scale)
    # the hidden-only form (what the fused spline kernels' callers use under autograd) has the net's own width
    monkeypatch.setattr(ResidualNet, "fuse_training", True)
    hidden = net.hidden(x)
    assert hidden.shape == (B, H) and hidden.requires_grad and len(calls) == 2
    monkeypatch.setattr(ResidualNet, "fuse_training", False)
    assert (hidden - net.hidden(x)).abs().max().item() <= 1e-5 * (1 + hidden.abs().max().item())


@pytest.mark.parametrize("di,nb,H", [(32, 2, 128), (12, 1, 128), (64, 3, 128), (36, 0, 128), (8, 2, 64), (20, 1, 52)])
def test_training_stream_packer_kernel_matches_the_tensor_reference(di, nb, H):
    """nfa_pack_resnet_hidden_train_f32 (one launch inside every training step) writes the bytes of the
    tensor-operation packer (whose layout the CPU suite decodes: tests/test_host_logic.py)."""
    from nflows_amd import ops
    net = _k14_net(128, di, nb, H)
    blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)
              for b in net.blocks]
    for final in (None, (net.final_layer.weight, net.final_layer.bias)):   # (40 outputs: one full tile + 8 rows)
        got = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks, final)
        ref = ops.pack_resnet_hidden_train_reference(net.initial_layer.weight, net.initial_layer.bias, blocks, final)
        assert got[0].shape == ref[0].shape and torch.equal(got[0].view(torch.int16), ref[0].view(torch.int16))
        assert torch.equal(got[1], ref[1])
        assert got[2].shape == ref[2].shape and torch.equal(got[2].view(torch.int16), ref[2].view(torch.int16))
        assert (got[3] is None and ref[3] is None) if final is None else torch.equal(got[3], ref[3])


@pytest.mark.parametrize("B,di,nb", [(65536, 32, 2), (16384, 64, 3), (2048, 8, 1)])
def test_fused_conditioner_training_kernels_at_size(B, di, nb):
    """The two K14 kernels at the benchmark's batch (every CU holds two workgroups), every array they write against
    float64 tensor operations: the forward arrays (hidden, relu(h_k), relu(a_k): continuous in the inputs) within
    2e-6 of the scale; the backward arrays with the ReLU masks PINNED to the ones the forward kernel produced (a
    float64 chain through the same masks: gradients are then continuous too) within 2e-6 of the scale; repeated
    launches bit-identical."""
    from nflows_amd import ops
    net = _k14_net(B, di, nb)
    x = torch.randn(B, di, device=DEV)
    g = torch.randn(B, 128, device=DEV)
    blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)
              for b in net.blocks]
    with torch.no_grad():
        final = (net.final_layer.weight, net.final_layer.bias)
        fw, fb, bw, fbias = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks, final)
        hid, saved, out = ops.resnet_hidden_forward(x, fw, fb, nb, fbias, 40)
        gx, grads = ops.resnet_hidden_backward(g, bw, saved, di)
        for _ in range(8):   # (fresh buffers every time: the kernels' correctness rests on counted waits)
            hid2, saved2, out2 = ops.resnet_hidden_forward(x, fw, fb, nb, fbias, 40)
            gx2, grads2 = ops.resnet_hidden_backward(g, bw, saved2, di)
            assert torch.equal(hid, hid2) and torch.equal(saved, saved2) and torch.equal(gx, gx2) and torch.equal(grads, grads2)
            assert torch.equal(out, out2)
        # the hidden-only form of the kernel (no final Linear in the stream) gives the same hidden activations
        fw0, fb0, _, _ = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks)
        hid0, saved0, none = ops.resnet_hidden_forward(x, fw0, fb0, nb)
        assert none is None and torch.equal(hid0, hid) and torch.equal(saved0, saved)
        d = lambda t: t.detach().double()
        h = torch.nn.functional.linear(d(x), d(net.initial_layer.weight), d(net.initial_layer.bias))
        truth_saved = []
        for w0, b0, w1, b1 in blocks:
            t = torch.relu(h)
            u = torch.relu(torch.nn.functional.linear(t, d(w0), d(b0)))
            truth_saved += [t, u]
            h = h + torch.nn.functional.linear(u, d(w1), d(b1))

        def close(name, got, truth):
            err, scale = (d(got) - truth).abs().max().item(), 1.0 + truth.abs().max().item()
            assert err <= 2e-6 * scale, "%s: %.3e (scale %.2e)" % (name, err, scale)

        close("hidden", hid, h)
        close("conditioner output", out, torch.nn.functional.linear(h, d(net.final_layer.weight), d(net.final_layer.bias)))
        for i in range(2 * nb):
            close("saved[%d]" % i, saved[i], truth_saved[i])
        gh = d(g)
        for k in reversed(range(nb)):
            w0, b0, w1, b1 = blocks[k]
            ga = (gh @ d(w1)) * (saved[2 * k + 1] > 0)
            gh = gh + (ga @ d(w0)) * (saved[2 * k] > 0)
            close("grads[%d]" % (2 * k + 1), grads[2 * k + 1], ga)
            close("grads[%d]" % (2 * k), grads[2 * k], gh)
        close("grad_inputs", gx, gh @ d(net.initial_layer.weight))


@pytest.mark.parametrize("B,di,nb,out", [(65536, 32, 2, 736), (16384, 64, 3, 40), (2048, 8, 1, 24), (1024, 12, 0, 368), (512, 32, 2, 4)])
def test_backward_kernel_from_the_output_gradient(B, di, nb, out):
    """nfa_resnet_backward_f32 (round 4): K14's backward kernel starting from d loss / d params -- the final Linear's
    input gradient as its first GEMM, g_params streamed through wave-private LDS images by LDS-DMA.  Widths: the
    benchmark's 736 (46 full k-steps), 40 and 24 and 368 (a last k-step of 8 columns: the clamped chunks meet zero
    weights), 4 (one quarter k-step).  Against float64 with the forward kernel's masks pinned, within 2e-6 of the scale;
    d loss / d hidden (written for K10) likewise; repeated launches bit-identical; and against the two-step route
    (library GEMM + nfa_resnet_hidden_backward_f32) within the same bound."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(B + out)
    net = ResidualNet(di, out, 128, num_blocks=nb).to(DEV)
    with torch.no_grad():
        for b in net.blocks:
            b.linear_layers[1].weight.mul_(40.0)
        net.final_layer.weight.mul_(30.0)
    x = torch.randn(B, di, device=DEV)
    gp = torch.randn(B, out, device=DEV) * torch.exp(2.0 * torch.randn(B, 1, device=DEV))   # (rows of very different scale)
    blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)
              for b in net.blocks]
    with torch.no_grad():
        final = (net.final_layer.weight, net.final_layer.bias)
        fw, fb, bw, fbias = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks, final)
        hid, saved, params = ops.resnet_hidden_forward(x, fw, fb, nb, fbias, out)
        gx, grads, ghid = ops.resnet_backward(gp, bw, saved, di)
        for _ in range(6):
            gx2, grads2, ghid2 = ops.resnet_backward(gp, bw, saved, di)
            assert torch.equal(gx, gx2) and torch.equal(grads, grads2) and torch.equal(ghid, ghid2)
        d = lambda t: t.detach().double()

        def close(name, got, truth):
            err, scale = (d(got) - truth).abs().max().item(), 1.0 + truth.abs().max().item()
            assert err <= 2e-6 * scale, "%s: %.3e (scale %.2e)" % (name, err, scale)

        gh = d(gp) @ d(net.final_layer.weight)
        close("grad_hidden", ghid, gh)
        for k in reversed(range(nb)):
            w0, b0, w1, b1 = blocks[k]
            ga = (gh @ d(w1)) * (saved[2 * k + 1] > 0)
            gh = gh + (ga @ d(w0)) * (saved[2 * k] > 0)
            close("grads[%d]" % (2 * k + 1), grads[2 * k + 1], ga)
            close("grads[%d]" % (2 * k), grads[2 * k], gh)
        close("grad_inputs", gx, gh @ d(net.initial_layer.weight))
        # the two-step route of round 3 on the same arrays
        gx3, grads3 = ops.resnet_hidden_backward(gp @ net.final_layer.weight, bw, saved, di)
        close("grad_inputs (two-step route)", gx3, gh @ d(net.initial_layer.weight))


@pytest.mark.parametrize("name", ["g_flow_nsf", "g_flow_nsf_h128"])
def test_flow_training_gradients_match_reference(G, name, monkeypatch):
    """loss = -mean log_prob through a whole flow: loss, input gradient and every parameter gradient against the
    REFERENCE's autograd (grads.npz).  The H = 128 instance on 128 rows runs the conditioners through K14 (counted)."""
    from nflows_amd import configs, ops
    calls = []
    real = ops.resnet_hidden_forward
    monkeypatch.setattr(ops, "resnet_hidden_forward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
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
    assert len(calls) == (2 * cfg["L"] if cfg["H"] == 128 else 0)


@pytest.mark.parametrize("B,D,K,tails,inverse,perm", [
    (257, 64, 8, "linear", False, "in"), (100, 64, 8, "linear", True, "out"), (33, 9, 5, "linear", False, None),
    (64, 12, 3, None, False, None), (64, 12, 10, None, True, None), (5, 784, 8, "linear", False, None),
    (3, 40, 16, "linear", True, "out"),
])
def test_rqs_coupling_gradients_vs_eager_autograd(B, D, K, tails, inverse, perm):
    """Fresh shapes (ragged batches, odd K, wide rows, fused permutations): HIP backward vs
    autograd through the CPU eager port in float64 (truth) and float32 (the reference's own path)."""
    _check_coupling_gradients(B, D, K, tails, inverse, perm, alternating=False)


@pytest.mark.parametrize("B,D,K,tails,inverse,perm", [
    (264, 64, 8, "linear", False, "in"), (259, 64, 8, "linear", True, "out"), (512, 64, 8, "linear", False, None),
    (96, 16, 8, None, False, "out"), (70, 16, 8, None, True, "in"), (1030, 128, 8, "linear", False, None),
])
def test_rqs_coupling_gradients_pipelined_kernel(B, D, K, tails, inverse, perm):
    """Alternating masks at K = 8 (d_t * P a multiple of 4): the software-pipelined backward kernel,
    with and without leftover rows behind the last full tile."""
    _check_coupling_gradients(B, D, K, tails, inverse, perm, alternating=True)


def _check_coupling_gradients(B, D, K, tails, inverse, perm, alternating):
    from nflows_amd import ops
    from oracle import eager
    rng = np.random.RandomState(B * 7 + D + K)
    mask = (np.arange(D) % 2 == 0) if alternating else rng.rand(D) < 0.5
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


def test_gradients_through_the_autoregressive_inverse():
    """Backward through `.inverse()` of the autoregressive layers (training an inverse
    autoregressive flow, reparameterised `sample_and_log_prob`): the column-wise inverse keeps the
    conditioner's saved inputs intact, and its parameter and input gradients equal those of the
    reference's out-of-place D-pass loop."""
    from nflows_amd.transforms import (InverseTransform, MaskedAffineAutoregressiveTransform,
                                       MaskedPiecewiseRationalQuadraticAutoregressiveTransform)
    torch.manual_seed(5)
    layers = [MaskedAffineAutoregressiveTransform(features=6, hidden_features=16),
              MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features=6, hidden_features=16, num_bins=4,
                                                                      tails="linear", tail_bound=3.0)]
    for layer in layers:
        layer = layer.to(DEV).train()
        z0 = torch.randn(64, 6, device=DEV)
        grads = []
        for columnwise in (True, False):
            layer.columnwise_inverse = columnwise
            layer.zero_grad()
            z = z0.clone().requires_grad_(True)
            x, lad = InverseTransform(layer)(z)          # = layer.inverse
            (x.pow(2).sum() + (lad * torch.arange(64, device=DEV)).sum()).backward()
            grads.append([z.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None])
        for a, b in zip(*grads):
            scale = 1 + b.abs().max().item()
            assert (a - b).abs().max().item() <= 2e-4 * scale
        assert len(grads[0]) > 3


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


def test_graphed_training_step_matches_eager_steps():
    """`GraphedTrainStep` (forward + backward + Adam in ONE HIP graph) follows the same loss
    trajectory as the eager loop from the same initial weights and batches."""
    import copy
    import nflows_amd
    from nflows_amd import configs
    from nflows_amd.graphs import GraphedTrainStep
    torch.manual_seed(0)
    flow_a = configs.rq_nsf_flow(num_layers=4, features=16, num_bins=8, hidden_features=32).to(DEV).train()
    flow_b = copy.deepcopy(flow_a)
    batches = [torch.randn(512, 16, device=DEV) * 0.8 + 0.2 for _ in range(6)]
    warm = 2

    opt_a = torch.optim.Adam(flow_a.parameters(), lr=1e-3, capturable=True)
    eager_losses = []
    for i, xb in enumerate([batches[0]] * warm + batches):  # the graphed step warms up on batches[0]
        opt_a.zero_grad(set_to_none=True)
        loss = -flow_a.log_prob(xb).mean()
        loss.backward()
        opt_a.step()
        if i >= warm:
            eager_losses.append(loss.item())

    opt_b = torch.optim.Adam(flow_b.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(flow_b, opt_b, batches[0], warmup=warm)
    graphed_losses = [step(xb).item() for xb in batches]
    nflows_amd.check_status()
    np.testing.assert_allclose(graphed_losses, eager_losses, rtol=2e-5, atol=2e-5)
    assert graphed_losses[-1] < graphed_losses[0]
    with pytest.raises(ValueError):
        step(batches[0][:100])
    with pytest.raises(ValueError):
        GraphedTrainStep(flow_b, torch.optim.Adam(flow_b.parameters(), lr=1e-3), batches[0])


@pytest.mark.parametrize("B,I,O", [
    (4096, 128, 128), (4100, 32, 128), (1000, 128, 736), (31, 128, 128), (2048, 64, 96), (5000, 200, 40),
    (64, 4, 4), (65536, 128, 128), (33, 16, 260),
])
def test_linear_wgrad_kernel(B, I, O):
    """K10 (weight / bias gradient of a conditioner layer, batch split over the chip) against the
    float64 product: at most 4x as far from it as the library's fp32 GEMM, and deterministic."""
    from nflows_amd import ops
    g = torch.Generator().manual_seed(B + I + O)
    x = torch.randn(B, I, generator=g)
    gy = torch.randn(B, O, generator=g) * torch.rand(1, O, generator=g)
    gw64 = (gy.double().t() @ x.double()).numpy()
    gb64 = gy.double().sum(0).numpy()
    xd, gyd = x.to(DEV), gy.to(DEV)
    gw, gb = ops.linear_wgrad(xd, gyd)
    close_to_truth(gw, (gyd.t() @ xd).cpu().numpy(), gw64, "grad_weight", tol=2e-6)
    close_to_truth(gb, gyd.sum(0).cpu().numpy(), gb64, "grad_bias", tol=2e-6)
    gw2, gb2 = ops.linear_wgrad(xd, gyd)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none = ops.linear_wgrad(xd, gyd, need_bias=False)
    assert none is None and torch.equal(gw3, gw)
    # a contiguous view that is not 16-byte aligned is copied; odd widths have no kernel
    flat = torch.cat([torch.zeros(1), x.reshape(-1)]).to(DEV)
    shifted = flat[1:].view(B, I)
    assert shifted.data_ptr() % 16 != 0
    assert torch.equal(ops.linear_wgrad(shifted, gyd)[0], gw)
    assert ops.linear_wgrad(torch.randn(B, I + 1, device=DEV), gyd) is None
    with pytest.raises(NotImplementedError):
        ops.linear_wgrad(x, gyd)


@pytest.mark.parametrize("B,I,O,count", [(65536, 128, 128, 4), (4100, 128, 128, 4), (8192, 32, 128, 2), (2048, 128, 736, 3),
                                          (37, 64, 64, 8), (4096, 128, 128, 1)])
def test_batched_linear_wgrad_kernel(B, I, O, count):
    """K10 for several same-shaped layers in one launch pair (round 4: the four hidden Linears of a conditioner):
    every problem against the float64 product with the single-problem rule (at most 4x as far from it as the
    library's fp32 GEMM), deterministic from call to call, per-problem bias switch, and within rounding of the
    single-problem entry point (another number of batch slices: other summation order)."""
    from nflows_amd import ops
    g = torch.Generator().manual_seed(B + I + O + count)
    probs = []
    for q in range(count):
        x = torch.randn(B, I, generator=g) * (0.5 + q)
        gy = torch.randn(B, O, generator=g) * torch.rand(1, O, generator=g)
        probs.append((x, gy))
    dev = [(x.to(DEV), gy.to(DEV)) for x, gy in probs]
    got = ops.linear_wgrad_batched(dev)
    again = ops.linear_wgrad_batched(dev)
    no_bias = ops.linear_wgrad_batched(dev, need_bias=False)
    for q, ((x, gy), (xd, gyd)) in enumerate(zip(probs, dev)):
        gw64 = (gy.double().t() @ x.double()).numpy()
        gb64 = gy.double().sum(0).numpy()
        close_to_truth(got[q][0], (gyd.t() @ xd).cpu().numpy(), gw64, "grad_weight %d" % q, tol=2e-6)
        close_to_truth(got[q][1], gyd.sum(0).cpu().numpy(), gb64, "grad_bias %d" % q, tol=2e-6)
        assert torch.equal(got[q][0], again[q][0]) and torch.equal(got[q][1], again[q][1])
        assert no_bias[q][1] is None and torch.equal(no_bias[q][0], got[q][0])
        single = ops.linear_wgrad(xd, gyd)
        scale = 1.0 + float(single[0].abs().max())
        assert float((single[0] - got[q][0]).abs().max()) <= 2e-5 * scale
    with pytest.raises(ValueError):
        ops.linear_wgrad_batched(dev + [(dev[0][0][:, :4], dev[0][1])])
    assert ops.linear_wgrad_batched([(torch.randn(B, I + 1, device=DEV), dev[0][1])]) is None


def test_conditioner_training_gradients_through_wgrad_kernel(monkeypatch):
    """ResidualNet / MLP / MADE under autograd: parameter and input gradients with K10 in the loop
    equal those of the plain library path."""
    import nflows_amd.nn.functional as NF
    from nflows_amd.nn.nets import MLP, ResidualNet
    from nflows_amd.transforms.made import MADE
    torch.manual_seed(3)
    nets = [ResidualNet(32, 736, 128, num_blocks=2), MLP([16], [32], [64, 64]), MADE(12, 32, output_multiplier=2, num_blocks=2)]
    for net in nets:
        net = net.to(DEV)
        x = torch.randn(3000, net.initial_layer.in_features if hasattr(net, "initial_layer") else (16 if isinstance(net, MLP) else 12), device=DEV)
        w = torch.randn(3000, 1, device=DEV)
        grads = {}
        for rows in (0, 1 << 40):
            monkeypatch.setattr(NF, "_WGRAD_MIN_ROWS", rows)
            net.zero_grad()
            xi = x.clone().requires_grad_(True)
            (net(xi) * w).sum().backward()
            grads[rows] = [xi.grad.clone()] + [p.grad.clone() for p in net.parameters()]
        for a, b in zip(grads[0], grads[1 << 40]):
            scale = 1.0 + b.abs().max().item()
            assert (a - b).abs().max().item() <= 2e-5 * scale


def test_sibling_spline_gradients_match_reference_autograd(golden_dir):
    """tests/golden/splines_lq_grads.npz: gradients of the linear / quadratic / cubic spline functionals
    (forward and inverse, constrained and with linear tails) from the reference's autograd."""
    from nflows_amd.transforms import splines
    G = np.load(os.path.join(golden_dir, "splines_lq_grads.npz"))
    fns = {"lin": splines.linear_spline, "ulin": splines.unconstrained_linear_spline,
           "quad": splines.quadratic_spline, "uquad": splines.unconstrained_quadratic_spline,
           "cub": splines.cubic_spline, "ucub": splines.unconstrained_cubic_spline}
    for name, kind, kw in G["meta"]:
        fn = fns[name.split("_")[0]]
        kwargs = parse_kwargs(kw)
        n_logits = {"linear": 1, "quadratic": 2, "cubic": 4}[kind]
        for inverse in (False, True):
            pre = "%s/%s" % (name, "inv_" if inverse else "")
            x = dev(G[name + "/x"]).requires_grad_(True)
            logits = [dev(G["%s/logits%d" % (name, i)]).requires_grad_(True) for i in range(n_logits)]
            y, lad = fn(x, *logits, inverse=inverse, **kwargs)
            ((y * dev(G[name + "/wy"])).sum() + (lad * dev(G[name + "/wl"])).sum()).backward()
            close_to_truth(x.grad, G[pre + "gx"], G[pre + "gx64"], pre + "gx")
            for i, t in enumerate(logits):
                close_to_truth(t.grad, G["%sglogits%d" % (pre, i)], G["%sglogits%d64" % (pre, i)], "%sglogits%d" % (pre, i))
    import nflows_amd
    nflows_amd.ops._status_word(torch.device(DEV)).zero_()


def test_sibling_spline_layers_train():
    """A coupling flow on piecewise-linear / -quadratic layers takes optimisation steps (the
    reference's training loop) and lowers its loss."""
    from nflows_amd import transforms as T
    from nflows_amd.distributions import StandardNormal
    from nflows_amd.flows import Flow
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.utils import torchutils
    torch.manual_seed(1)
    for cls, kw in ((T.PiecewiseLinearCouplingTransform, dict(num_bins=8, tails="linear", tail_bound=4.0)),
                    (T.PiecewiseQuadraticCouplingTransform, dict(num_bins=8, tails="linear", tail_bound=4.0)),
                    (T.PiecewiseCubicCouplingTransform, dict(num_bins=8, tails="linear", tail_bound=4.0))):
        layers = []
        for i in range(2):
            layers.append(cls(torchutils.create_alternating_binary_mask(6, even=(i % 2 == 0)),
                              lambda a, b: ResidualNet(a, b, 32, num_blocks=1), **kw))
        flow = Flow(T.CompositeTransform(layers), StandardNormal([6])).to(DEV).train()
        opt = torch.optim.Adam(flow.parameters(), lr=5e-3)
        data = torch.randn(2048, 6, device=DEV) * 0.6 + 0.4
        losses = []
        for _ in range(40):
            opt.zero_grad()
            loss = -flow.log_prob(data).mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.05, losses[::10]
