#!/usr/bin/env python3
"""Debug aid: K8h (f16x2) against the bf16x3 engine on one layer, with parts of the conditioner
switched off (no residual blocks; block weights / biases zeroed), to find which part differs."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nflows_amd
from nflows_amd import configs
from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ

def run(tag, nb, edit=None):
    flow = configs.rq_nsf_flow(num_layers=1, features=64, num_bins=8, hidden_features=128, num_blocks=nb, seed=0)
    net = flow._transform._transforms[1].transform_net
    with torch.no_grad():
        if edit:
            edit(net)
    flow = flow.cuda().eval()
    x = torch.randn(8192, 64, generator=torch.Generator().manual_seed(1)).cuda()
    out = {}
    for engine in ("bf16x3", "f16x2"):
        RQ.conditioner_engine = engine
        with torch.no_grad():
            z, lad = flow._transform(x)
        out[engine] = z.cpu().numpy()
    d = np.abs(out["bf16x3"] - out["f16x2"])
    print("%-46s max|diff| %.3e  mean %.3e" % (tag, d.max(), d.mean()))

def zero_block_weights(net):
    for b in net.blocks:
        for l in b.linear_layers:
            l.weight.zero_()
def zero_block_w1(net):
    for b in net.blocks:
        b.linear_layers[1].weight.zero_(); b.linear_layers[1].bias.zero_()
def zero_block_w0(net):
    for b in net.blocks:
        b.linear_layers[0].weight.zero_()
def zero_biases(net):
    for n, p in net.named_parameters():
        if n.endswith("bias"):
            p.zero_()
def zero_final_w(net):
    net.final_layer.weight.zero_()
def big_init_bias(net):
    net.initial_layer.weight.zero_()

run("no blocks", 0)
run("no blocks, biases zero", 0, zero_biases)
run("no blocks, final weights zero", 0, zero_final_w)
run("no blocks, initial weights zero", 0, big_init_bias)
run("1 block", 1)
run("1 block, block weights zero", 1, zero_block_weights)
run("1 block, second Linear zero (w, b)", 1, zero_block_w1)
run("1 block, first Linear weight zero", 1, zero_block_w0)
run("1 block, biases zero", 1, zero_biases)
run("2 blocks", 2)
run("2 blocks, block weights zero", 2, zero_block_weights)
