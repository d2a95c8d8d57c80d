#!/usr/bin/env python3
"""K14 bring-up aid: every array the two training kernels write against eager fp32 tensor ops, with the rows that
differ.  python tools/k14_debug.py [B] [d_i] [blocks] [repeats]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
from nflows_amd.nn.nets import ResidualNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
di = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = "cuda:0"
torch.manual_seed(0)
net = ResidualNet(di, 40, 128, num_blocks=nb).to(dev)
with torch.no_grad():
    for b in net.blocks:
        b.linear_layers[1].weight.mul_(60.0)
        b.linear_layers[1].bias.mul_(60.0)
x = torch.randn(B, di, device=dev)
g = torch.randn(B, 128, device=dev)
blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias) for b in net.blocks]
fw, fb, bw, _ = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks)
with torch.no_grad():
    # eager reference
    h = torch.nn.functional.linear(x, net.initial_layer.weight, net.initial_layer.bias)
    ref_saved = []
    for w0, b0, w1, b1 in blocks:
        t = torch.relu(h)
        u = torch.relu(torch.nn.functional.linear(t, w0, b0))
        ref_saved += [t, u]
        h = h + torch.nn.functional.linear(u, w1, b1)
    ref_h = h
    gh = g.clone()
    ref_grads = [None] * (2 * nb)
    for k in reversed(range(nb)):
        w0, b0, w1, b1 = blocks[k]
        ga = (gh @ w1) * (ref_saved[2 * k + 1] > 0)
        gh = gh + (ga @ w0) * (ref_saved[2 * k] > 0)
        ref_grads[2 * k + 1] = ga
        ref_grads[2 * k] = gh
    ref_gx = gh @ net.initial_layer.weight

    def report(name, got, ref):
        err = (got - ref).abs()
        bad = (err > 1e-4 * (1 + ref.abs().max())).any(dim=1).nonzero().flatten()
        msg = "%-12s max err %.3e (scale %.2e)" % (name, err.max().item(), ref.abs().max().item())
        if bad.numel():
            q = bad // 128
            msg += "  BAD rows %d: first %s | quads %s | row%%128 in [%d, %d] | waves %s" % (
                bad.numel(), bad[:6].tolist(), torch.unique(q)[:8].tolist(), (bad % 128).min().item(), (bad % 128).max().item(),
                torch.unique((bad % 128) // 32).tolist())
        print(msg, flush=True)

    for rep in range(reps):
        print("--- repeat", rep)
        hid, saved, _ = ops.resnet_hidden_forward(x, fw, fb, nb)
        report("hidden", hid, ref_h)
        for i in range(2 * nb):
            report("saved[%d]" % i, saved[i], ref_saved[i])
        gx, grads = ops.resnet_hidden_backward(g, bw, torch.stack(ref_saved) if nb else saved, di)
        for i in range(2 * nb):
            report("grads[%d]" % i, grads[i], ref_grads[i])
        report("grad_x", gx, ref_gx)
