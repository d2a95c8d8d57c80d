#!/usr/bin/env python3
"""K14's two kernels as the training step launches them (forward with the final Linear; backward from d loss / d params)
at B rows for one library build: NFLOWS_AMD_LIB=build_variants/<v>.so python tools/k14_stage_probe.py [B]
(HIP-graph replay, median of 20; stages per row block: forward 2 + 16 + 46, backward 46 + 16 + 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nflows_amd import ops
from nflows_amd.nn.nets import ResidualNet
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
torch.manual_seed(0)
net = ResidualNet(32, 736, 128, num_blocks=2).to(dev)
x = torch.randn(B, 32, device=dev)
gp = torch.randn(B, 736, device=dev)
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
    fw, fb, bw, fbias = ops.pack_resnet_hidden_train(net.initial_layer.weight, net.initial_layer.bias, blocks,
                                                     (net.final_layer.weight, net.final_layer.bias))
    hid, saved, _ = ops.resnet_hidden_forward(x, fw, fb, 2, fbias, 736)
    tag = os.path.basename(os.environ.get("NFLOWS_AMD_LIB", "product"))
    t_f = timeit(lambda: ops.resnet_hidden_forward(x, fw, fb, 2, fbias, 736))
    t_b = timeit(lambda: ops.resnet_backward(gp, bw, saved, 32))
    t_h = timeit(lambda: ops.resnet_hidden_backward(g, bw, saved, 32))
    print("%s rows %d: forward + final %.1f us (%.2f us / stage) | backward from params %.1f us (%.2f us / stage) | backward from hidden %.1f us"
          % (tag, B, t_f, t_f / 64, t_b, t_b / 64, t_h), flush=True)
