#!/usr/bin/env python3
"""GPU time of the pieces of K14's training path for one conditioner (ResidualNet 32 -> 128 x 2 blocks) at B rows:
the weight packer, the forward kernel, the backward kernel, the five K10 weight-gradient calls -- beside the eager
modules' forward and backward (library GEMMs + elementwise kernels).  HIP-graph replay, median of 20.
    python tools/k14_micro.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
from nflows_amd.nn.nets import ResidualNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
torch.manual_seed(0)
net = ResidualNet(32, 736, 128, num_blocks=2).to(dev)
x = torch.randn(B, 32, device=dev)
g = torch.randn(B, 128, device=dev)
blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias) for b in net.blocks]


def timeit(fn, reps=20, inner=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(inner):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); graph.replay(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3 / inner


with torch.no_grad():
    pack = lambda: ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks)
    fw, fb, bw, _ = pack()
    hid, saved, _ = ops.resnet_hidden_forward(x, fw, fb, 2)
    gx, grads = ops.resnet_hidden_backward(g, bw, saved, 32)
    print("rows %d" % B)
    print("packer (torch ops)            %8.1f us" % timeit(pack))
    print("K14 forward kernel            %8.1f us" % timeit(lambda: ops.resnet_hidden_forward(x, fw, fb, 2)))
    print("K14 backward kernel           %8.1f us" % timeit(lambda: ops.resnet_hidden_backward(g, bw, saved, 32)))
    fw2, fb2, _, fbias = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks,
                                                      (net.final_layer.weight, net.final_layer.bias))
    print("K14 forward + final Linear    %8.1f us" % timeit(lambda: ops.resnet_hidden_forward(x, fw2, fb2, 2, fbias, 736)))
    print("library final Linear 128->736 %8.1f us" % timeit(lambda: torch.nn.functional.linear(hid, net.final_layer.weight, net.final_layer.bias)))

    def wgrads():
        ops.linear_wgrad(x, grads[0])
        ops.linear_wgrad(saved[0], grads[1]); ops.linear_wgrad(saved[1], grads[2])
        ops.linear_wgrad(saved[2], grads[3]); ops.linear_wgrad(saved[3], g)
    print("K10 x 5 (weight gradients)    %8.1f us" % timeit(wgrads))

# eager modules: forward, and forward + backward of the hidden part
ResidualNet.fuse_training = False
xg = x.clone().requires_grad_(True)


def eager_fb():
    net.zero_grad(set_to_none=True)
    xg.grad = None
    h = net.hidden(xg)
    h.backward(g)


with torch.no_grad():
    print("eager hidden forward (no grad)%8.1f us" % timeit(lambda: net.hidden(x)))
for _ in range(3):
    eager_fb()
torch.cuda.synchronize()
evs = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); eager_fb(); e.record(); evs.append((s, e))
torch.cuda.synchronize()
print("eager hidden fwd + bwd (autograd, incl. K10), launched from the host %8.1f us" % (sorted(a.elapsed_time(b) for a, b in evs)[10] * 1e3))
ResidualNet.fuse_training = True


def fused_fb():
    net.zero_grad(set_to_none=True)
    xg.grad = None
    h = net.hidden(xg)
    h.backward(g)


for _ in range(3):
    fused_fb()
torch.cuda.synchronize()
evs = []
for _ in range(20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fused_fb(); e.record(); evs.append((s, e))
torch.cuda.synchronize()
print("K14 hidden fwd + bwd (autograd, incl. packer and K10), from the host %8.1f us" % (sorted(a.elapsed_time(b) for a, b in evs)[10] * 1e3))
