"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the drop-in classes keep the reference's constructor / buffer / state_dict contract,
argument errors match the reference's, and the product never falls back to a CPU path."""
import ctypes
import os
import re

import math
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    header = open(os.path.join(ROOT, "include", "nflows_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(nfa_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 11
    from nflows_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libnflows_amd.so does not export " + name
    assert sorted(_native.EXPORTS) == declared  # the Python binding covers the whole ABI
    loaded = _native.load()
    assert loaded.nfa_abi_version() == _native.ABI_VERSION
    assert loaded.nfa_build_arch() == b"gfx950"
    assert loaded.nfa_strerror(_native.ERR_MIN_BIN_WIDTH) == b"Minimal bin width too large for the number of bins"


def test_header_constants_match_binding():
    header = open(os.path.join(ROOT, "include", "nflows_amd.h")).read()
    from nflows_amd import _native as N
    consts = dict(re.findall(r"#define\s+(NFA_[A-Z0-9_]+)\s+(\d+)", header))
    assert int(consts["NFA_ABI_VERSION"]) == N.ABI_VERSION
    for py, c in [("OK", "NFA_OK"), ("ERR_INVALID_ARGUMENT", "NFA_ERR_INVALID_ARGUMENT"),
                  ("ERR_UNSUPPORTED", "NFA_ERR_UNSUPPORTED"), ("ERR_MIN_BIN_WIDTH", "NFA_ERR_MIN_BIN_WIDTH"),
                  ("ERR_MIN_BIN_HEIGHT", "NFA_ERR_MIN_BIN_HEIGHT"), ("ERR_HIP", "NFA_ERR_HIP"),
                  ("STATUS_OUTSIDE_DOMAIN", "NFA_STATUS_OUTSIDE_DOMAIN"),
                  ("STATUS_NEG_DISCRIMINANT", "NFA_STATUS_NEG_DISCRIMINANT"),
                  ("STATUS_BAD_INDEX", "NFA_STATUS_BAD_INDEX"), ("TAILS_NONE", "NFA_TAILS_NONE"),
                  ("FLAG_INVERSE", "NFA_FLAG_INVERSE"), ("FLAG_ACCUMULATE_LOGABSDET", "NFA_FLAG_ACCUMULATE_LOGABSDET"),
                  ("FLAG_WEIGHTS_BF16X3", "NFA_FLAG_WEIGHTS_BF16X3"), ("FLAG_LOGITS_LOG2E", "NFA_FLAG_LOGITS_LOG2E"),
                  ("TAILS_LINEAR", "NFA_TAILS_LINEAR"), ("SCALE_DEFAULT", "NFA_SCALE_DEFAULT"),
                  ("SCALE_GENERAL", "NFA_SCALE_GENERAL"), ("SCALE_ADDITIVE", "NFA_SCALE_ADDITIVE"),
                  ("SCALE_GIVEN", "NFA_SCALE_GIVEN"), ("SCALE_SOFTPLUS", "NFA_SCALE_SOFTPLUS"),
                  ("FLAG_STANDARD_NORMAL_LOG_PROB", "NFA_FLAG_STANDARD_NORMAL_LOG_PROB"),
                  ("FLAG_SKIP_OUTPUTS", "NFA_FLAG_SKIP_OUTPUTS"),
                  ("FLAG_PAD_COLUMNS_SHIFT", "NFA_FLAG_PAD_COLUMNS_SHIFT")]:
        assert getattr(N, py) == int(consts[c]), c
    # struct layout: 2 int32 + 10 doubles, natural alignment
    assert ctypes.sizeof(N.RqsSpec) == 8 + 10 * 8


def test_spec_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it can be exercised on CPU through the
    C ABI with NULL pointers and batch 0 / bad specs."""
    from nflows_amd import _native as N
    from nflows_amd import ops
    lib = N.load()
    with pytest.raises(ValueError, match="Minimal bin width too large"):
        ops.make_rqs_spec(4, "linear", min_bin_width=0.3)
    with pytest.raises(ValueError, match="Minimal bin height too large"):
        ops.make_rqs_spec(4, None, min_bin_height=0.3)
    with pytest.raises(RuntimeError, match="cubic tails are not implemented"):
        ops.make_rqs_spec(4, "cubic")
    spec = ops.make_rqs_spec(8, "linear", tail_bound=3.0)
    assert spec.left == -3.0 and spec.top == 3.0 and spec.tails == N.TAILS_LINEAR
    assert abs(spec.tail_logit - np.log(np.exp(1 - 1e-3) - 1)) == 0.0
    # batch == 0 is a valid no-op; negative sizes and bad enums are rejected
    null = None
    assert lib.nfa_rqs_coupling_f32(null, null, null, null, null, null, null, null, null, 0, 64, 32,
                                    ctypes.byref(spec), 0, null) == N.OK
    assert lib.nfa_rqs_coupling_f32(null, null, null, null, null, null, null, null, null, -1, 64, 32,
                                    ctypes.byref(spec), 0, null) == N.ERR_INVALID_ARGUMENT
    assert lib.nfa_rqs_coupling_f32(null, null, null, null, null, null, null, null, null, 4, 64, 65,
                                    ctypes.byref(spec), 0, null) == N.ERR_INVALID_ARGUMENT
    assert lib.nfa_rqs_coupling_f32(null, null, null, null, null, null, null, null, null, 4, 64, 32,
                                    ctypes.byref(spec), 0, null) == N.ERR_INVALID_ARGUMENT  # NULL data
    bad = ops.make_rqs_spec(8, "linear")
    bad.min_bin_width = 0.5
    assert lib.nfa_rqs_coupling_f32(null, null, null, null, null, null, null, null, null, 0, 64, 32,
                                    ctypes.byref(bad), 0, null) == N.ERR_MIN_BIN_WIDTH
    assert lib.nfa_affine_coupling_f32(null, null, null, null, null, null, null, null, null, 4, 8, 4,
                                       99, 0, null) == N.ERR_INVALID_ARGUMENT
    assert lib.nfa_rowsum_f32(null, null, 0, 5, null) == N.OK
    assert lib.nfa_searchsorted_f32(null, 0, 10, null, null, 0, 1e-6, null) == N.OK
    assert lib.nfa_searchsorted_f32(null, 0, 0, null, null, 4, 1e-6, null) == N.ERR_INVALID_ARGUMENT
    assert lib.nfa_searchsorted_f32(null, 0, 10, null, null, 4, 1e-6, null) == N.ERR_INVALID_ARGUMENT  # NULL data
    assert lib.nfa_permute_cols_b32(null, null, null, null, 3, 0, null) == N.ERR_INVALID_ARGUMENT


def test_no_cpu_fallback():
    from nflows_amd import configs, ops
    from nflows_amd.transforms import RandomPermutation, splines
    flow = configs.rq_nsf_flow(num_layers=1, features=8, num_bins=4, hidden_features=16)
    x = torch.randn(4, 8)
    with torch.no_grad():
        with pytest.raises(NotImplementedError, match="no CPU fallback"):
            flow.log_prob(x)
        with pytest.raises(NotImplementedError, match="no CPU fallback"):
            RandomPermutation(8)(x)
        with pytest.raises(NotImplementedError, match="no CPU fallback"):
            splines.unconstrained_rational_quadratic_spline(x[:, 0], torch.zeros(4, 4), torch.zeros(4, 4),
                                                            torch.zeros(4, 3))
        with pytest.raises(NotImplementedError):
            ops.rowsum(x)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no file of the package mentions it."""
    pkg = os.path.join(ROOT, "nflows_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "nfa_oracle" not in text and "oracle/" not in text and "oracle." not in text, f
    # measurement scripts under tools/ do not use it either (those that do live under tests/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f


def test_state_dict_contract(golden_dir):
    """Same buffer / parameter names as the reference: its state_dicts load strictly."""
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, "flows.npz"))
    flow = configs.rq_nsf_flow(num_layers=2, features=64, num_bins=8, hidden_features=32)
    want = sorted(k[len("nsf_d64/sd/"):] for k in g.files if k.startswith("nsf_d64/sd/"))
    assert sorted(flow.state_dict().keys()) == want
    assert "_distribution._log_z" not in flow.state_dict()  # non-persistent, like the reference
    flow = configs.moons_maf_flow()
    want = sorted(k[len("moons_maf/sd/"):] for k in g.files if k.startswith("moons_maf/sd/"))
    assert sorted(flow.state_dict().keys()) == want
    flow = configs.ar_rq_flow(12, 32, 8, 3.0, 2)
    want = sorted(k[len("ar_rq_small/sd/"):] for k in g.files if k.startswith("ar_rq_small/sd/"))
    assert sorted(flow.state_dict().keys()) == want


def test_constructor_contract_and_errors():
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import (AdditiveCouplingTransform, AffineCouplingTransform, Permutation,
                                       PiecewiseRationalQuadraticCouplingTransform, RandomPermutation,
                                       ReversePermutation, Transform, InverseNotAvailable)
    from nflows_amd.utils import create_alternating_binary_mask, create_mid_split_binary_mask

    def net(i, o):
        return ResidualNet(i, o, hidden_features=16)
    t = PiecewiseRationalQuadraticCouplingTransform(create_alternating_binary_mask(7), net, num_bins=4,
                                                    tails="linear", tail_bound=2.0)
    assert t.features == 7 and t.num_transform_features == 4 and t.num_identity_features == 3
    assert t.transform_features.tolist() == [0, 2, 4, 6] and t.identity_features.tolist() == [1, 3, 5]
    assert t.transform_features.dtype == torch.int64
    assert t._transform_dim_multiplier() == 11 and t.transform_net.final_layer.out_features == 44
    assert (t.num_bins, t.tails, t.tail_bound, t.min_bin_width) == (4, "linear", 2.0, 1e-3)
    t2 = PiecewiseRationalQuadraticCouplingTransform(create_mid_split_binary_mask(6), net, num_bins=4)
    assert t2._transform_dim_multiplier() == 13 and t2.tails is None
    a = AffineCouplingTransform([1, -1, 1, -1], net)
    assert a._transform_dim_multiplier() == 2 and a.transform_features.tolist() == [0, 2]
    assert a.scale_activation is AffineCouplingTransform.DEFAULT_SCALE_ACTIVATION
    assert AdditiveCouplingTransform([1, 0], net)._transform_dim_multiplier() == 1
    with pytest.raises(ValueError, match="Mask must be a 1-dim tensor"):
        AffineCouplingTransform(torch.ones(2, 2), net)
    with pytest.raises(ValueError, match="Mask can't be empty"):
        AffineCouplingTransform(torch.ones(0), net)
    with pytest.raises(ValueError, match="Permutation must be a 1D tensor"):
        Permutation(torch.zeros(2, 2, dtype=torch.long))
    with pytest.raises(ValueError, match="dim must be a positive integer"):
        Permutation(torch.arange(3), dim=0)
    with pytest.raises(ValueError):
        RandomPermutation(0)
    assert ReversePermutation(4)._permutation.tolist() == [3, 2, 1, 0]
    p = RandomPermutation(16)
    assert sorted(p._permutation.tolist()) == list(range(16))
    assert torch.equal(p._inverse_permutation[p._permutation], torch.arange(16))
    with pytest.raises(InverseNotAvailable):
        Transform().inverse(torch.zeros(1))
    with pytest.raises(ValueError, match="Expected features = 7, got 5"):
        t(torch.zeros(3, 5))
    with pytest.raises(ValueError, match="Inputs must be a 2D or a 4D tensor"):
        t(torch.zeros(3))


def test_masks_and_leading_dim_helpers(golden_dir):
    from nflows_amd.utils import torchutils as tu
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    for f in (7, 8, 64):
        assert np.array_equal(tu.create_alternating_binary_mask(f, True).numpy(), g["mask_alt_even_%d" % f])
        assert np.array_equal(tu.create_alternating_binary_mask(f, False).numpy(), g["mask_alt_odd_%d" % f])
        assert np.array_equal(tu.create_mid_split_binary_mask(f).numpy(), g["mask_mid_%d" % f])
    m = tu.create_random_binary_mask(9)
    assert m.dtype == torch.uint8 and int(m.sum()) == 5
    x = torch.arange(24).reshape(6, 4)
    assert tu.split_leading_dim(x, [2, 3]).shape == (2, 3, 4)
    assert tu.merge_leading_dims(tu.split_leading_dim(x, [2, 3]), 2).equal(x)
    assert tu.repeat_rows(x[:2], 3).tolist() == [x[0].tolist()] * 3 + [x[1].tolist()] * 3
    with pytest.raises(TypeError):
        tu.repeat_rows(x, 0)
    with pytest.raises(ValueError):
        tu.merge_leading_dims(x, 3)
    g2 = np.load(os.path.join(golden_dir, "searchsorted.npz"))
    for which in ("left", "right", "mid"):
        idx = tu.searchsorted(torch.from_numpy(g2["knots"])[None, :], torch.from_numpy(g2[which + "_in"]))
        assert idx.tolist() == list(range(9))


def test_distribution_template_methods():
    from nflows_amd.distributions import Distribution, StandardNormal
    d = StandardNormal([3])
    assert d.sample(5).shape == (5, 3)
    assert d.sample(7, batch_size=3).shape == (7, 3)
    assert d.sample(2, context=torch.zeros(4, 1)).shape == (4, 2, 3)
    assert d.mean().tolist() == [0.0, 0.0, 0.0]
    with pytest.raises(TypeError):
        d.sample(0)
    with pytest.raises(ValueError):
        d.log_prob(torch.zeros(2, 3), context=torch.zeros(3, 1))
    with pytest.raises(RuntimeError):
        Distribution()(1)
    assert "_log_z" not in d.state_dict() and d._log_z.dtype == torch.float64


@pytest.mark.parametrize("K", [8, 10, 4, 12, 16, 24])
def test_whole_layer_packing_is_a_lossless_rearrangement(K):
    """Host side of K8 (ops.pack_resnet_conditioner, runs on CPU tensors): emulate what the kernel
    computes with the packed weights -- every GEMM transposed, the k index permuted the way the
    accumulator layout of the previous layer dictates, three bf16 pieces per weight -- and compare
    with the PyTorch network in float64.  Checks the stage order, the column / row permutations,
    the bias order and the folded 1/sqrt(hidden) scale without a GPU."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(0)
    dt, di, P = 8, 6, 3 * K - 1
    R = ops.final_rows_per_feature(P)               # rows per feature after padding: 24 (8 bins), else whole 16s
    assert R == {8: 24, 10: 32, 4: 16, 12: 48, 16: 48, 24: 80}[K]
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=2).double()
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn_like(p_) * 0.3)
    wp, bp = ops.pack_resnet_conditioner(net.float(), dt, P)
    net = net.double()
    tiles = dt * R // 32
    assert wp.shape == (2 + 16 * 2 + 2 * tiles, 768 * 8) and wp.dtype == torch.bfloat16
    assert bp.shape == (128 + 256 * 2 + tiles * 32,)
    w = wp.double().view(-1, 768, 8)          # [stage][vec4 slot][8 bf16]
    x = torch.randn(32, di, dtype=torch.float64)    # one wave's 32 samples
    lane_r = torch.arange(64) % 32
    lane_h = torch.arange(64) // 32

    def acc_to_features(acc):   # acc[t][lane][q] -> [sample, feature]
        out = torch.zeros(32, 32 * acc.shape[0], dtype=torch.float64)
        for t in range(acc.shape[0]):
            for q in range(16):
                feat = 32 * t + 8 * (q // 4) + 4 * lane_h + q % 4
                out[lane_r, feat] = acc[t, :, q]
        return out

    def bias_tiles(off, n):     # [n tiles][2 halves][16] -> acc[t][lane][q]
        b = bp[off:off + n * 32].double().view(n, 2, 16)
        return b[:, lane_h, :].clone()

    def mfma(acc_t, a_frag, b_frag):
        # a_frag[lane][8]: weights of out-row (lane % 32), k = (lane // 32, j); b_frag[lane][8]: acts of
        # sample (lane % 32); D[i][n] = sum_k A[i][k] B[k][n]; lane l holds column n = l % 32, rows by q
        A = torch.zeros(32, 16, dtype=torch.float64)
        Bm = torch.zeros(16, 32, dtype=torch.float64)
        for l in range(64):
            A[l % 32, 8 * (l // 32):8 * (l // 32) + 8] = a_frag[l]
            Bm[8 * (l // 32):8 * (l // 32) + 8, l % 32] = b_frag[l]
        Dm = A @ Bm
        for l in range(64):
            for q in range(16):
                acc_t[l, q] += Dm[8 * (q // 4) + 4 * (l // 32) + q % 4, l % 32]

    def pieces_sum(stage, base):   # hi + mid + lo of the A fragment starting at vec4 slot `base`
        return w[stage, base:base + 64] + w[stage, base + 64:base + 128] + w[stage, base + 128:base + 192]

    # activations as the B operand of k-step ks: lane holds 8 values
    def b_from_acc(acc, ks):       # previous layer's accumulators -> [lane][8]
        return acc[ks // 2][:, 8 * (ks % 2):8 * (ks % 2) + 8]

    stage = 0
    bx = torch.zeros(2, 64, 8, dtype=torch.float64)
    for ks in range(2):
        for l in range(64):
            for j in range(8):
                i = ks * 16 + (l // 32) * 8 + j
                bx[ks, l, j] = x[l % 32, i] if i < di else 0.0
    h = bias_tiles(0, 4)
    for ks in range(2):                              # initial layer: k-major [4 tiles][3 pieces][64]
        for t in range(4):
            mfma(h[t], pieces_sum(stage, (t * 3) * 64), bx[ks])
        stage += 1
    boff = 128
    for blk in range(2):
        for which in range(2):
            src = torch.relu(h) if which == 0 else u
            acc = bias_tiles(boff, 4)
            if which == 1:
                acc = acc + h
            for ks in range(8):
                for t in range(4):
                    mfma(acc[t], pieces_sum(stage, (t * 3) * 64), b_from_acc(src, ks))
                stage += 1
            boff += 128
            if which == 0:
                u = torch.relu(acc)
            else:
                h = acc
    got_hidden = acc_to_features(h)
    want_hidden = net.hidden(x)
    assert (got_hidden - want_hidden).abs().max().item() < 2e-6 * want_hidden.abs().max().item()  # bf16x3 weights
    # final layer: tile-major, two stages per tile [3 pieces][4 k-steps][64]; rows in K7 order
    out = torch.zeros(tiles, 64, 16, dtype=torch.float64)
    bfin = bias_tiles(boff, tiles)
    for t in range(tiles):
        out[t] = bfin[t]
        for hs in range(2):
            for k4 in range(4):
                a_frag = sum(w[stage, (p_ * 4 + k4) * 64:(p_ * 4 + k4) * 64 + 64] for p_ in range(3))
                mfma(out[t], a_frag, b_from_acc(h, hs * 4 + k4))
            stage += 1
    assert stage == wp.shape[0]
    want = net.final_layer(want_hidden).view(32, dt, P).clone()
    want[..., :2 * K] /= np.sqrt(128.0)              # the folded 1/sqrt(hidden) scale
    if K != 8:  # T tiles per group (10 bins: two): the 16 T values of a lane-half are feature 2g + half
        T = R // 16
        for g in range(dt // 2):
            for half in range(2):
                lanes = torch.arange(32) + 32 * half
                vals = torch.cat([out[T * g + t][lanes] for t in range(T)], dim=1)   # [32 samples][16 T]
                ref = want[:, 2 * g + half]
                assert (vals[:, :P] - ref).abs().max().item() < 5e-6 * (1 + ref.abs().max().item()), (g, half)
                assert not vals[:, P:].any()                                          # the pad rows
        return
    for g in range(dt // 4):
        for half in range(2):
            lanes = torch.arange(32) + 32 * half
            vals = torch.cat([out[3 * g + t][lanes] for t in range(3)], dim=1)   # [32 samples][48]
            for f in range(2):
                feature = 4 * g + 2 * half + f
                got = vals[:, 24 * f:24 * f + 23]
                ref = want[:, feature]
                assert (got - ref).abs().max().item() < 5e-6 * (1 + ref.abs().max().item()), (g, half, f)
                assert vals[:, 24 * f + 23].abs().max().item() == 0.0              # the pad row


@pytest.mark.parametrize("K", [3, 5, 11, 13, 32])
def test_f16x3_packing_of_the_other_bin_counts(K):
    """K8x's final layer at bin counts other than 8 (round 6): K8h's general row rule -- a feature's 3 K - 1 logits padded to
    16 T rows, group g's T tiles hand lane-half h the logits of feature 2 g + h (ops._k8_row_order_32) -- in K8x's
    tile-major 12 KB stages.  The f16 fragments of every (tile, k-step), taken back to rows, are the final Linear's weights
    (x T, the width / height rows / sqrt(hidden)) to the two pieces' 2^-21; bias and stage counts are the kernel's."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(K)
    dt, di, P = 8, 8, 3 * K - 1
    R = ops.final_rows_per_feature(P)
    assert R == 16 * ((P + 15) // 16)
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=1).float()
    wp, bp, sc = ops.pack_resnet_conditioner_f16x3(net, dt, P)
    tiles = dt * R // 32
    assert wp.shape == (2 + 16 + 2 * tiles, 768 * 8) and bp.shape == (128 + 256 + tiles * 32,) and sc.shape == (8,)
    kappa, ST = float(sc[6]), float(sc[7])
    assert kappa * ST == 1.0 and math.frexp(ST)[0] == 0.5
    T = ST / ops.K8X_ACT_SCALE
    raw = wp.view(torch.uint8).view(-1, 12, 1024)[2 + 16:]             # the final layer's stages
    scale = torch.ones(P, dtype=torch.float64)
    scale[:2 * K] = 1.0 / math.sqrt(128)
    want = net.final_layer.weight.detach().double().view(dt, P, 128) * scale[None, :, None] * T
    bias = net.final_layer.bias.detach().double().view(dt, P) * scale[None, :] * ST
    order_k = ops._k8_column_order()
    tiles_per_group = R // 16
    for t in range(tiles):
        rows = torch.zeros(32, 128, dtype=torch.float64)
        for ks in range(8):
            st, k = 2 * t + ks // 4, ks % 4
            hfrag = raw[st, 4 * (k // 2) + 2 * (k % 2)].view(torch.float16).view(2, 32, 8).double()      # [half][row][8]
            lfrag = raw[st, 4 * (k // 2) + 2 * (k % 2) + 1].view(torch.float16).view(2, 32, 8).double()
            for hf in range(2):
                rows[:, order_k[ks * 16 + hf * 8:ks * 16 + hf * 8 + 8]] = (hfrag + lfrag)[hf]
        g, tg = t // tiles_per_group, t % tiles_per_group
        for i in range(32):                      # tile row i = 8 (q // 4) + 4 half + q % 4 -> logit 16 tg + q of feature 2 g + half
            half, q = (i >> 2) & 1, ((i >> 3) << 2) | (i & 3)
            logit = 16 * tg + q
            ref = want[2 * g + half, logit] if logit < P else torch.zeros(128, dtype=torch.float64)
            assert (rows[i] - ref).abs().max().item() <= 2.0 ** -20 * max(1.0, float(ref.abs().max())), (t, i)
            got_b = float(bp[128 + 256 + t * 32 + half * 16 + q])
            ref_b = float(bias[2 * g + half, logit]) if logit < P else 0.0
            assert abs(got_b - ref_b) <= 1e-6 * max(1.0, abs(ref_b)), (t, i)


def test_whole_layer_packing_takes_the_tails_none_parameter_count():
    """tails=None (round 6): P = 3 K + 1 logits per feature; ops.pack_resnet_conditioner pads them to whole 16-row shares in
    the general row order (K = 8: 25 -> 32 rows, where linear tails use the 24-row order), width / height rows divided by
    sqrt(hidden), the K + 1 derivative rows untouched; the eligibility rules keep the f16 engines and padded geometries away."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from nflows_amd.utils import torchutils
    torch.manual_seed(3)
    for K, dt in ((8, 8), (10, 4), (4, 12)):
        P = 3 * K + 1
        R = ops.final_rows_per_feature(P)
        assert R == 16 * ((P + 15) // 16)
        net = ResidualNet(6, dt * P, hidden_features=128, num_blocks=1).float()
        wp, bp = ops.pack_resnet_conditioner(net, dt, P)
        assert wp.shape[0] == 2 + 16 + 2 * (dt * R // 32) and bp.shape == (128 + 256 + dt * R,)
        # the biases of feature 0 (lane-half 0 of the group's tiles): width / height entries scaled, derivative entries not
        got = bp[128 + 256:].view(-1, 2, 16)[0:R // 16, 0, :].reshape(-1)[:P]
        want = net.final_layer.bias.detach().view(dt, P)[0].clone()
        want[:2 * K] /= math.sqrt(128)
        assert torch.allclose(got, want, atol=1e-7), K
    layer = RQ(torchutils.create_alternating_binary_mask(16, even=True), lambda i, o: ResidualNet(i, o, hidden_features=128, num_blocks=2),
               num_bins=10, tails=None)
    assert layer._transform_dim_multiplier() == 31 and not layer._use_f16() and not layer._use_f16x3() and not layer._log2e()
    assert layer._run_signature()[5:7] == (None, (8, 8))
    odd = RQ(torchutils.create_alternating_binary_mask(10, even=True), lambda i, o: ResidualNet(i, o, hidden_features=128, num_blocks=2),
             num_bins=10, tails=None)
    with torch.no_grad():
        assert not odd._resnet_eligible(None)      # five transformed features: a spare column would be needed


@pytest.mark.parametrize("dt", [32, 12, 4])
def test_f16_colsplit_packing_is_the_tile16_packing_rearranged(dt):
    """Host side of K8c (ops.pack_resnet_conditioner_f16(tile16=True, colsplit=True), round 6): parameter words and every
    stage in front of the final layer are K8s's; the final layer's fragment pairs are K8s's, regrouped -- pair 2 w + q of
    stage (round r, tile i, half s) = K8s's pair of tile 6 (4 r + w) + i, k-step 2 s + q --, zero fragments for the groups a
    wave has none of in the last round."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(dt)
    di, P, blocks = 20, 23, 2
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=blocks).float()
    ws, prm_s = ops.pack_resnet_conditioner_f16(net, dt, P, tile16=True)
    wc, prm_c = ops.pack_resnet_conditioner_f16(net, dt, P, tile16=True, colsplit=True)
    assert torch.equal(prm_s, prm_c)
    head = 1 + 8 * blocks                       # one 32-wide k-step of the initial layer, four per hidden Linear
    groups, rounds = dt // 4, (dt // 4 + 3) // 4
    assert ws.shape == (head + groups * 3, 8192) and wc.shape == (head + rounds * 12, 8192)
    assert torch.equal(ws[:head], wc[:head])
    fs = ws[head:].view(groups * 3, 8, 1024)    # [stage = tile pair][pair = 4 (tile % 2) + k-step][hi | lo: 512 f16 each]
    fc = wc[head:].view(rounds, 6, 2, 4, 2, 1024)   # [r][i][s][w][q]
    for r in range(rounds):
        for w in range(4):
            G = 4 * r + w
            for i in range(6):
                for S in range(4):
                    got = fc[r, i, S // 2, w, S % 2]
                    if G >= groups:
                        assert not got.view(torch.int16).any()
                        continue
                    tile = 6 * G + i
                    assert torch.equal(got.view(torch.int16), fs[tile // 2, 4 * (tile % 2) + S].view(torch.int16)), (r, w, i, S)


@pytest.mark.parametrize("di", [6, 40])
def test_f16x3_whole_layer_packing_carries_the_scales(di):
    """Host side of K8x (ops.pack_resnet_conditioner_f16x3, round 6): emulate csrc/rqs_resnet_f16x3.hip's data flow on one
    32-row tile from the packed blobs -- 12 KB stages of twelve fragments (k-major: two k-steps of two tiles, [H0, L0, H1, L1,
    X lo, X hi] per tile; final layer: four k-steps of a tile, [H0, L0, H1, L1][H2, L2, H3, L3][X01][X23]), the three f16 products of every
    k-step, the bf8 instruction of every pair of k-steps on the packed bytes and the high bytes of the activation pieces (x
    the 2^-8 block scale), activations as three f16 pieces at scale S with the last one kept x 2^8 (what split3_scaled
    makes), accumulators at S T, the residual stream rebuilt from its pieces x T, the {1 / T, T} pairs, logits =
    accumulators x kappa -- and compare with the PyTorch network in float64."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(0)
    dt, K = 8, 8
    P = 23
    S = ops.K8X_ACT_SCALE
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=2).double()
    with torch.no_grad():
        for i_, p_ in enumerate(net.parameters()):
            p_.copy_(torch.randn_like(p_) * (0.3 if i_ % 3 else 0.004))   # GEMMs of very different magnitudes
    wp, bp, sc = ops.pack_resnet_conditioner_f16x3(net.float(), dt, P)
    net = net.double()
    init_ks = 4 if di > 32 else 2
    tiles = dt * 24 // 32
    assert wp.shape == (init_ks + 16 * 2 + 2 * tiles, 768 * 8) and wp.dtype == torch.float16
    assert bp.shape == (128 + 256 * 2 + tiles * 32,) and sc.shape == (2 * 6,) and sc.dtype == torch.float32
    for g in range(6):   # powers of two, {1 / T, T} (final: {1 / (S T), S T})
        assert math.frexp(float(sc[2 * g]))[0] == 0.5 and float(sc[2 * g]) * float(sc[2 * g + 1]) == 1.0
    raw = wp.view(torch.uint8).view(-1, 12, 1024)                      # [stage][fragment][64 lanes x 16 B]
    x = torch.randn(32, di, dtype=torch.float64).float().double()    # one wave's 32 samples (fp32 values)
    lane_r = torch.arange(64) % 32
    lane_h = torch.arange(64) // 32

    def f16_frag(stage, f):      # -> [64 lanes][8] float64
        return raw[stage, f].view(torch.float16).view(64, 8).double()

    def x_bytes(stage, f):       # fragments f, f + 1 -> [64 lanes][32] float64 (bf8 decoded)
        b = torch.cat((raw[stage, f].view(64, 16), raw[stage, f + 1].view(64, 16)), dim=1)
        return b.view(torch.float8_e5m2).float().double()

    def bf8_trunc(v16):          # an f16 tensor's high bytes, decoded
        return (v16.view(torch.int16) & -256).view(torch.float16).double()

    def split3(v):   # what split3_scaled does to (already scaled) fp32 values: hi, lo, r' = RN16(256 d) as f16 tensors
        v = v.float()
        hi = v.to(torch.float16)
        t = v - hi.float()
        lo = t.to(torch.float16)
        r = ((t - lo.float()) * 256.0).to(torch.float16)
        return hi, lo, r

    def value(pc):
        return pc[0].double() + pc[1].double() + pc[2].double() / 256.0

    def acc_to_features(acc):   # acc[t][lane][q] -> [sample, feature]
        out = torch.zeros(32, 32 * acc.shape[0], dtype=torch.float64)
        for t in range(acc.shape[0]):
            for q in range(16):
                out[lane_r, 32 * t + 8 * (q // 4) + 4 * lane_h + q % 4] = acc[t, :, q]
        return out

    def bias_tiles(off, n):     # [n tiles][2 halves][16] -> acc[t][lane][q]
        return bp[off:off + n * 32].double().view(n, 2, 16)[:, lane_h, :].clone()

    def mfma(acc_t, a_frag, b_frag, kpl):   # kpl: k values per lane of this instruction (8: f16 32x32x16, 32: bf8 32x32x64)
        A = torch.zeros(32, 2 * kpl, dtype=torch.float64)
        Bm = torch.zeros(2 * kpl, 32, dtype=torch.float64)
        for l in range(64):
            A[l % 32, kpl * (l // 32):kpl * (l // 32) + kpl] = a_frag[l]
            Bm[kpl * (l // 32):kpl * (l // 32) + kpl, l % 32] = b_frag[l]
        Dm = A @ Bm
        for l in range(64):
            for q in range(16):
                acc_t[l, q] += Dm[8 * (q // 4) + 4 * (l // 32) + q % 4, l % 32]

    def pair_products(acc_t, stage, base, b0, b1):
        """NFA_K8X_PAIR: fragments base .. base + 5 = H0, L0, H1, L1, X lo, X hi; b0 / b1: (hi, lo, r') [64][8] f16 of the two k-steps"""
        for k, b in ((0, b0), (1, b1)):
            ah, al = f16_frag(stage, base + 2 * k), f16_frag(stage, base + 2 * k + 1)
            mfma(acc_t, ah, b[1].double(), 8)
            mfma(acc_t, al, b[0].double(), 8)
            mfma(acc_t, ah, b[0].double(), 8)
        bx = torch.cat((bf8_trunc(b0[2]), bf8_trunc(b0[0]), bf8_trunc(b1[2]), bf8_trunc(b1[0])), dim=1)   # [64][32]
        small = torch.zeros(64, 16, dtype=torch.float64)
        mfma(small, x_bytes(stage, base + 4), bx, 32)
        acc_t += small / 256.0

    def b_from_acc(pieces, ks):       # pieces of the previous layer's accumulators -> three [lane][8]
        return tuple(pc[ks // 2][:, 8 * (ks % 2):8 * (ks % 2) + 8] for pc in pieces)

    def pieces_of(acc, scale, relu):  # [4][64][16] accumulators x scale -> three piece tensors
        v = acc * scale
        if relu:
            v = torch.relu(v)
        return split3(v)

    def gemm_kmajor(acc, stage, nks, b_of):
        for pr in range(nks // 2):
            b0, b1 = b_of(2 * pr), b_of(2 * pr + 1)
            for half in range(2):
                pair_products(acc[2 * half], stage, 0, b0, b1)
                pair_products(acc[2 * half + 1], stage, 6, b0, b1)
                stage += 1
        return stage

    stage = 0
    bx0 = torch.zeros(init_ks, 64, 8, dtype=torch.float64)
    for ks in range(init_ks):
        for l in range(64):
            for j in range(8):
                i = ks * 16 + (l // 32) * 8 + j
                bx0[ks, l, j] = x[l % 32, i] if i < di else 0.0
    xp = split3(bx0 * S)
    big = (bx0 * S).abs() >= 2.0 ** -9
    assert torch.equal(value(xp)[big], (bx0 * S)[big])                 # all 24 bits while the last piece (x 2^8) is >= 2^-24 ...
    assert (value(xp) - bx0 * S).abs().max().item() <= 2.0 ** -33      # ... and an absolute 2^-33 (/ S) below that
    acc = bias_tiles(0, 4)
    stage = gemm_kmajor(acc, stage, init_ks, lambda ks: tuple(pc[ks] for pc in xp))
    g = 0
    h = pieces_of(acc.float().double(), float(sc[0]), False)     # S h as pieces (accumulators are fp32 on the device)
    boff = 128
    g += 1
    for blk in range(2):
        keep = (h[0].double() >= 0) | torch.isnan(h[0].double())
        relu_h = tuple(torch.where(keep, pc, torch.zeros_like(pc)) for pc in h)   # the sign mask on the pieces
        acc = bias_tiles(boff, 4)
        stage = gemm_kmajor(acc, stage, 8, lambda ks: b_from_acc(relu_h, ks))
        skip = bias_tiles(boff + 128, 4) + value(h) * float(sc[2 * (g + 1) + 1])     # skip: pieces x T (before u is converted)
        u = pieces_of(acc.float().double(), float(sc[2 * g]), True)
        g += 1
        acc = skip.float().double()
        stage = gemm_kmajor(acc, stage, 8, lambda ks: b_from_acc(u, ks))
        h = pieces_of(acc.float().double(), float(sc[2 * g]), False)
        g += 1
        boff += 256
    got_hidden = acc_to_features(value(h)) / S
    want_hidden = net.hidden(x)
    assert (got_hidden - want_hidden).abs().max().item() < 1e-6 * want_hidden.abs().max().item()
    out = torch.zeros(tiles, 64, 16, dtype=torch.float64)
    bfin = bias_tiles(boff, tiles)
    for t in range(tiles):                                # final layer: two stages per tile, [H0, L0, H1, L1][H2, L2, H3, L3][X01][X23]
        out[t] = bfin[t]
        for hs in range(2):
            for j in range(2):
                k0 = hs * 4 + 2 * j
                b0, b1 = b_from_acc(h, k0), b_from_acc(h, k0 + 1)
                for k, b in ((2 * j, b0), (2 * j + 1, b1)):
                    ah, al = f16_frag(stage, 2 * k), f16_frag(stage, 2 * k + 1)
                    mfma(out[t], ah, b[1].double(), 8)
                    mfma(out[t], al, b[0].double(), 8)
                    mfma(out[t], ah, b[0].double(), 8)
                bx = torch.cat((bf8_trunc(b0[2]), bf8_trunc(b0[0]), bf8_trunc(b1[2]), bf8_trunc(b1[0])), dim=1)
                small = torch.zeros(64, 16, dtype=torch.float64)
                mfma(small, x_bytes(stage, 8 + 2 * j), bx, 32)
                out[t] += small / 256.0
            stage += 1
    assert stage == wp.shape[0]
    kappa = float(sc[2 * g])
    assert kappa * float(sc[2 * g + 1]) == 1.0
    # the diagnostic twins' packed order and its inverse (ops.unpack_last_layer_logits)
    packed = torch.zeros(32, tiles * 32)
    for t in range(tiles):
        for l in range(64):
            packed[l % 32, 32 * t + 16 * (l // 32):32 * t + 16 * (l // 32) + 16] = (out[t, l] * kappa).float()
    logits = ops.unpack_last_layer_logits(packed, dt).double()
    want = net.final_layer(want_hidden).view(32, dt, P).clone()
    want[..., :2 * K] /= np.sqrt(128.0)
    assert (logits - want).abs().max().item() < 2e-6 * (1 + want.abs().max().item())


def _emulate_f16_whole_layer(wp, prm, x, di, dt, K, num_blocks, ctx=None):
    """The data flow of K8h on ONE 32-row tile, from the packed stream: every GEMM on two f16 weight pieces
    pre-scaled by a power of two, accumulator <-> piece conversions with the headers' scales, the fp32 residual
    stream, and -- with `ctx` [32, ce] -- the context k-steps of the initial layer and the gate of every block.
    Returns (hidden activations x S as [32, 128], logits as [32, dt, 3K - 1] (widths / heights pre-divided by
    sqrt(hidden)), stages consumed, parameter words consumed)."""
    H = 4  # header floats
    P = 3 * K - 1
    R = 24 if K == 8 else 16 * ((P + 15) // 16)
    tiles = dt * R // 32
    w = wp.double().view(-1, 1024, 8)
    bp = prm
    lane_r = torch.arange(64) % 32
    lane_h = torch.arange(64) // 32

    def acc_to_features(acc):
        out = torch.zeros(32, 32 * acc.shape[0], dtype=torch.float64)
        for t in range(acc.shape[0]):
            for q in range(16):
                out[lane_r, 32 * t + 8 * (q // 4) + 4 * lane_h + q % 4] = acc[t, :, q]
        return out

    def bias_tiles(off, n):
        return bp[off:off + n * 32].double().view(n, 2, 16)[:, lane_h, :].clone()

    def mfma(acc_t, a_frag, b_frag):
        A = torch.zeros(32, 16, dtype=torch.float64)
        Bm = torch.zeros(16, 32, dtype=torch.float64)
        for l in range(64):
            A[l % 32, 8 * (l // 32):8 * (l // 32) + 8] = a_frag[l]
            Bm[8 * (l // 32):8 * (l // 32) + 8, l % 32] = b_frag[l]
        Dm = A @ Bm
        for l in range(64):
            for q in range(16):
                acc_t[l, q] += Dm[8 * (q // 4) + 4 * (l // 32) + q % 4, l % 32]

    def pair(stage, g):       # fragment pair g of a stage: hi + lo
        return w[stage, (2 * g) * 64:(2 * g) * 64 + 64] + w[stage, (2 * g + 1) * 64:(2 * g + 1) * 64 + 64]

    def b_from_acc(acc, ks):  # pieces of k-step ks = tile ks // 2, registers 8 (ks % 2) ..
        return acc[ks // 2][:, 8 * (ks % 2):8 * (ks % 2) + 8]

    def b_from_rows(rows, ks):  # pieces of k-step ks of an input given by rows: k = ks*16 + half*8 + j
        b = torch.zeros(64, 8, dtype=torch.float64)
        for l in range(64):
            for j in range(8):
                i = ks * 16 + (l // 32) * 8 + j
                b[l, j] = rows[l % 32, i] if i < rows.shape[1] else 0.0
        return b

    def k_major(stage0, acc, pieces_of):  # a stage per two k-steps, pair 4 (ks % 2) + t = tile t
        st = stage0
        for ks, b in enumerate(pieces_of):
            for t in range(4):
                mfma(acc[t], pair(st, 4 * (ks % 2) + t), b)
            st += ks % 2
        return st

    def tile_major(stage0, acc, src):   # the final GEMM: one stage per tile, pair ks = k-step ks
        st = stage0
        for t in range(acc.shape[0]):
            for ks in range(8):
                mfma(acc[t], pair(st, ks), b_from_acc(src, ks))
            st += 1
        return st

    stage, off = 0, 0
    if ctx is None:
        init_pieces = [b_from_rows(x[:, :di], ks) for ks in range(4 if di > 32 else 2)]
    else:  # [identity features, zero-padded to 32 | context, zero-padded to 32]
        cpieces = [b_from_rows(ctx, ks) for ks in range(2)]
        init_pieces = [b_from_rows(x[:, :di], ks) for ks in range(2)] + cpieces
    out_scale = bp[off].double()
    hacc = bias_tiles(off + H, 4)                     # the fp32 residual stream
    stage = k_major(stage, hacc, init_pieces)
    hp = torch.relu(hacc * out_scale)                 # pieces of relu(h) at scale S
    off += H + 128
    for blk in range(num_blocks):
        out_scale = bp[off].double()
        assert bp[off + 1] == 0
        u = bias_tiles(off + H, 4)
        stage = k_major(stage, u, [b_from_acc(hp, ks) for ks in range(8)])
        q = torch.relu(u * out_scale)
        off += H + 128
        out_scale, ratio = bp[off].double(), bp[off + 1].double()
        if ctx is None:
            hacc = hacc * ratio + bias_tiles(off + H, 4)  # skip connection: the stream itself, rescaled
            stage = k_major(stage, hacc, [b_from_acc(q, ks) for ks in range(8)])
            off += H + 128
        else:
            v = bias_tiles(off + H, 4)                    # the second Linear alone ...
            stage = k_major(stage, v, [b_from_acc(q, ks) for ks in range(8)])
            off += H + 128
            inv_t = bp[off].double()
            gate = bias_tiles(off + H, 4)                 # ... the gate's Linear on the context pieces
            stage = k_major(stage, gate, cpieces)
            hacc = hacc * ratio + v * torch.sigmoid(gate * inv_t)
            off += H + 128
        hp = hacc * out_scale
        if blk + 1 < num_blocks:
            hp = torch.relu(hp)
    hidden = acc_to_features(hp)
    kappa, inv_kappa = bp[off].double(), bp[off + 1].double()
    assert kappa * inv_kappa == 1.0
    out = bias_tiles(off + H, tiles)
    stage = tile_major(stage, out, hp)
    off += H + tiles * 32
    out = out * kappa
    logits = torch.zeros(32, dt, P, dtype=torch.float64)
    if K != 8:   # T tiles per group of two features: lane-half h of group g holds feature 2g + h
        T = R // 16
        for g in range(dt // 2):
            for half in range(2):
                lanes = torch.arange(32) + 32 * half
                vals = torch.cat([out[T * g + t][lanes] for t in range(T)], dim=1)
                logits[:, 2 * g + half] = vals[:, :P]
                assert not vals[:, P:].any()   # (11 bins: 32 logits, no pad row)
        return hidden, logits, stage, off
    for g in range(dt // 4):
        for half in range(2):
            lanes = torch.arange(32) + 32 * half
            vals = torch.cat([out[3 * g + t][lanes] for t in range(3)], dim=1)
            for f in range(2):
                logits[:, 4 * g + 2 * half + f] = vals[:, 24 * f:24 * f + 23]
                assert vals[:, 24 * f + 23].abs().max().item() == 0.0
    return hidden, logits, stage, off


def _emulate_f16s_whole_layer(wp, prm, x, di, dt, num_blocks):
    """The data flow of K8s (csrc/rqs_resnet_f16s.hip) on ONE 16-row tile, from the packed stream: lane l = (sample
    n = l % 16, lane group g = l // 16); v_mfma_f32_16x16x32_f16 with A[m = l % 16][k = 8 g + j], B[k = 8 g + j][n],
    D[4 g + i][n]; a k-major stage = the eight output tiles of one 32-wide k-step, a final-layer stage = two 16-row
    tiles over four k-steps; biases in natural row order; the six tiles of a group of four features give lane group g
    the 24 logits of feature 4 G + g.  Returns (hidden x S [16, 128], logits [16, dt, 23], stages, words consumed)."""
    H, P = 4, 23
    w = wp.double().view(-1, 1024, 8)        # [stage][16 x 64 lanes][8 halves]
    bp = prm
    ln = torch.arange(64) % 16
    lg = torch.arange(64) // 16

    def pair(stage, t):       # fragment pair t of a stage (hi + lo): [64 lanes, 8]
        return w[stage, (2 * t) * 64:(2 * t) * 64 + 64] + w[stage, (2 * t + 1) * 64:(2 * t + 1) * 64 + 64]

    def mfma(acc_t, a_frag, b_frag):   # acc_t [64 lanes, 4]
        A = torch.zeros(16, 32, dtype=torch.float64)
        Bm = torch.zeros(32, 16, dtype=torch.float64)
        for l in range(64):
            A[l % 16, 8 * (l // 16):8 * (l // 16) + 8] = a_frag[l]
            Bm[8 * (l // 16):8 * (l // 16) + 8, l % 16] = b_frag[l]
        Dm = A @ Bm
        for i in range(4):
            acc_t[:, i] += Dm[4 * lg + i, ln]

    def bias_tiles(off, n):   # [n tiles][64 lanes][4]: rows 4 g + i of every 16-row tile, natural order
        return bp[off:off + n * 16].double().view(n, 4, 4)[:, lg, :].clone()

    def b_from_acc(acc, S):   # pieces of k-step S: tiles 2 S (j < 4) and 2 S + 1 (j >= 4)
        return torch.cat((acc[2 * S], acc[2 * S + 1]), dim=1)

    def b_from_rows(rows, S):  # k = 32 S + 8 g + j
        b = torch.zeros(64, 8, dtype=torch.float64)
        for l in range(64):
            for j in range(8):
                i = 32 * S + 8 * (l // 16) + j
                b[l, j] = rows[l % 16, i] if i < rows.shape[1] else 0.0
        return b

    def k_major(stage0, acc, pieces_of):
        st = stage0
        for b in pieces_of:
            for t in range(8):
                mfma(acc[t], pair(st, t), b)
            st += 1
        return st

    stage, off = 0, 0
    out_scale = bp[off].double()
    hacc = bias_tiles(off + H, 8)
    stage = k_major(stage, hacc, [b_from_rows(x[:, :di], S) for S in range(2 if di > 32 else 1)])
    hp = torch.relu(hacc * out_scale)
    off += H + 128
    for blk in range(num_blocks):
        out_scale = bp[off].double()
        u = bias_tiles(off + H, 8)
        stage = k_major(stage, u, [b_from_acc(hp, S) for S in range(4)])
        q = torch.relu(u * out_scale)
        off += H + 128
        out_scale, ratio = bp[off].double(), bp[off + 1].double()
        hacc = hacc * ratio + bias_tiles(off + H, 8)
        stage = k_major(stage, hacc, [b_from_acc(q, S) for S in range(4)])
        off += H + 128
        hp = hacc * out_scale
        if blk + 1 < num_blocks:
            hp = torch.relu(hp)
    hidden = torch.zeros(16, 128, dtype=torch.float64)
    for t in range(8):
        for i in range(4):
            hidden[ln, 16 * t + 4 * lg + i] = hp[t][:, i]
    kappa, inv_kappa = bp[off].double(), bp[off + 1].double()
    assert kappa * inv_kappa == 1.0
    tiles = dt * 24 // 16
    out = bias_tiles(off + H, tiles)
    src = [b_from_acc(hp, S) for S in range(4)]
    for t in range(tiles):     # two tiles per stage: pairs 0..3 / 4..7 = the four k-steps
        for S in range(4):
            mfma(out[t], pair(stage + t // 2, 4 * (t % 2) + S), src[S])
    stage += tiles // 2
    off += H + tiles * 16
    out = out * kappa
    logits = torch.zeros(16, dt, P, dtype=torch.float64)
    for G in range(dt // 4):
        vals = torch.cat([out[6 * G + tau] for tau in range(6)], dim=1)    # [64 lanes, 24]
        for l in range(64):
            logits[l % 16, 4 * G + l // 16] = vals[l, :23]
            assert vals[l, 23].abs().item() == 0.0
    return hidden, logits, stage, off


@pytest.mark.parametrize("di", [6, 40])
def test_f16_tile16_packing_reproduces_the_network(di):
    """Host side of K8s (ops.pack_resnet_conditioner_f16(tile16=True)): the same stages and parameter words as K8h's
    stream, with the fragment / column / row / bias orders of the 16x16x32 tile -- emulate the kernel's data flow and
    compare with the PyTorch network in float64."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(2)
    K, dt = 8, 8
    P = 3 * K - 1
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=2).double()
    with torch.no_grad():
        for i_, p_ in enumerate(net.parameters()):
            p_.copy_(torch.randn_like(p_) * (0.3 if i_ % 3 else 0.004))
    wp, prm = ops.pack_resnet_conditioner_f16(net.float(), dt, P, tile16=True)
    wp32, prm32 = ops.pack_resnet_conditioner_f16(net.float(), dt, P)
    net = net.double()
    assert wp.shape == wp32.shape and prm.shape == prm32.shape        # one stream format for both tile shapes
    x = torch.randn(16, di, dtype=torch.float64)
    hidden, logits, stages, words_used = _emulate_f16s_whole_layer(wp, prm, x, di, dt, 2)
    assert stages == wp.shape[0] and words_used == prm.numel()
    want_hidden = net.hidden(x)
    assert (hidden - want_hidden).abs().max().item() < 1e-6 * want_hidden.abs().max().item()
    want = net.final_layer(want_hidden).view(16, dt, P).clone()
    want[..., :2 * K] /= np.sqrt(128.0)
    assert (logits - want).abs().max().item() < 2e-6 * (1 + want.abs().max().item())


@pytest.mark.parametrize("act_scale", [1.0, 16.0])
def test_f16_whole_layer_packing_carries_the_scales(act_scale, K=8):
    """Host side of K8h (ops.pack_resnet_conditioner_f16 / build_f16_stream): emulate the kernel's
    data flow with the packed stream -- the parameter stage (tables, per-GEMM headers {out_scale,
    skip_scale}, pre-scaled biases), every GEMM tile-major on two f16 weight pieces pre-scaled by a
    power of two, pieces of the activations at scale S, the residual stream kept in fp32 at the
    scale of the GEMM that wrote it, the final layer's logits = accumulators x kappa -- and compare
    with the PyTorch network in float64."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(0)
    dt, di = 8, 6
    P = 3 * K - 1
    net = ResidualNet(di, dt * P, hidden_features=128, num_blocks=2).double()
    with torch.no_grad():
        for i_, p_ in enumerate(net.parameters()):
            p_.copy_(torch.randn_like(p_) * (0.3 if i_ % 3 else 0.004))   # GEMMs of very different magnitudes
    wp, prm = ops.pack_resnet_conditioner_f16(net.float(), dt, P, act_scale=act_scale)
    net = net.double()
    tiles = dt * ops.final_rows_per_feature(P) // 32
    H = 4  # header floats
    assert wp.shape == (1 + 8 * 2 + tiles, 1024 * 8) and wp.dtype == torch.float16   # 16 KB stages of eight pairs
    assert prm.shape == ((H + 128) * 5 + H + tiles * 32,)
    assert torch.isfinite(wp.float()).all() and wp.float().abs().max() < 2 ** 14
    # the stream of a one-layer run: parameter stage + weight stages
    tables = torch.arange(256, dtype=torch.int32)
    stream, pstages, final = ops.build_f16_stream([(wp, prm)], tables)
    assert pstages == 1 and stream.shape == (1 + wp.shape[0], 8192) and torch.equal(final, tables[128:])
    words = stream[0].view(torch.float32)
    assert torch.equal(words[:128].view(torch.int32), tables[:128]) and torch.equal(words[128:128 + prm.numel()], prm)
    assert torch.equal(stream[1:], wp)
    x = torch.randn(32, di, dtype=torch.float64)
    hidden, logits, stages, words_used = _emulate_f16_whole_layer(wp, prm, x, di, dt, K, 2)
    assert stages == wp.shape[0] and words_used == prm.numel()
    want_hidden = net.hidden(x)
    assert (hidden / act_scale - want_hidden).abs().max().item() < 1e-6 * want_hidden.abs().max().item()
    want = net.final_layer(want_hidden).view(32, dt, P).clone()
    want[..., :2 * K] /= np.sqrt(128.0)
    assert (logits - want).abs().max().item() < 2e-6 * (1 + want.abs().max().item())


@pytest.mark.parametrize("K", [2, 4, 5, 7, 10, 11, 12, 16, 20, 32])
def test_f16_whole_layer_packing_of_other_bin_counts(K):
    """Round 4: the final layer of 2 .. 16 bins -- 3 K - 1 logits per feature padded to whole lane-half shares of 16,
    T = 1 .. 3 tiles per group of two features -- through the same emulation of K8h's data flow."""
    test_f16_whole_layer_packing_carries_the_scales(1.0, K)


def test_f16_whole_layer_packing_with_a_context():
    """The same for a conditioner with a context (nn/nets/resnet.py:9-52, :92-100): the context columns are
    the initial layer's last two k-steps, every block carries one more stage (the gate's Linear, two k-steps)
    and one more parameter group {1 / T_c, 0} + biases x T_c, and the residual stream takes
    (second Linear) x sigmoid(gate) in, rescaled by the second Linear's skip_scale."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(1)
    K, dt, di, ce = 8, 8, 11, 12
    P = 3 * K - 1
    net = ResidualNet(di, dt * P, hidden_features=96, context_features=ce, num_blocks=2).double()
    with torch.no_grad():
        for i_, p_ in enumerate(net.parameters()):
            p_.copy_(torch.randn_like(p_) * (0.3 if i_ % 3 else 0.02))
    wp, prm = ops.pack_resnet_conditioner_f16(net.float(), dt, P)
    net = net.double()
    tiles = dt * 24 // 32
    assert wp.shape == (2 + 9 * 2 + tiles, 1024 * 8)       # four initial k-steps, nine stages per block
    assert prm.shape == ((4 + 128) * 7 + 4 + tiles * 32,)
    x = torch.randn(32, di, dtype=torch.float64)
    ctx = torch.randn(32, ce, dtype=torch.float64)
    hidden, logits, stages, words_used = _emulate_f16_whole_layer(wp, prm, x, di, dt, K, 2, ctx=ctx)
    assert stages == wp.shape[0] and words_used == prm.numel()
    want = net(x, context=ctx).view(32, dt, P).clone()
    want[..., :2 * K] /= np.sqrt(96.0)
    assert (logits - want).abs().max().item() < 2e-6 * (1 + want.abs().max().item())
    with pytest.raises(ValueError):   # more than 32 identity features beside a context: the bf16x3 kernel's case
        ops.pack_resnet_conditioner_f16(ResidualNet(40, dt * P, hidden_features=64, context_features=4), dt, P)


def test_batch_norm_fold_is_the_same_network():
    """ops.fold_batch_norm: a ResidualNet with eval-mode batch norm in its blocks (resnet.py:24-27, :41-47) equals,
    in float64, the plain residual chain over the folded layers -- the stream carrying x + c_0 / a_0 in front of
    every block; no fold in training mode, with a non-positive gain in front of the ReLU, or with a context."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(0)
    net = ResidualNet(20, 37, hidden_features=48, num_blocks=3, use_batch_norm=True, dropout_probability=0.3).double()
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(48, generator=gen, dtype=torch.float64))
                m.running_var.copy_(torch.rand(48, generator=gen, dtype=torch.float64) * 3 + 0.1)
                m.weight.copy_(torch.rand(48, generator=gen, dtype=torch.float64) * 2 + 0.05)
                m.bias.copy_(torch.randn(48, generator=gen, dtype=torch.float64))
        for p in net.blocks[1].linear_layers[1].parameters():
            p.mul_(500.0)
    assert ops.fold_batch_norm(net) is None            # training mode: batch statistics
    net.eval()
    assert ops.fold_batch_norm(ResidualNet(4, 4, hidden_features=8)) is not None
    plain = ResidualNet(4, 4, hidden_features=8)
    assert ops.fold_batch_norm(plain) is plain
    x = torch.randn(257, 20, generator=gen, dtype=torch.float64)
    with torch.no_grad():
        want = net(x)
        # the folded layers are float32 (what the packers read): fold a float64 copy by hand for the identity ...
        folded = ops.fold_batch_norm(net)
        assert folded is not None and folded.context_features is None and len(folded.blocks) == 3
        s = torch.nn.functional.linear(x, folded.initial_layer.weight.double(), folded.initial_layer.bias.double())
        for b in folded.blocks:
            h = torch.relu(s)
            h = torch.nn.functional.linear(h, b.linear_layers[0].weight.double(), b.linear_layers[0].bias.double())
            h = torch.relu(h)
            s = s + torch.nn.functional.linear(h, b.linear_layers[1].weight.double(), b.linear_layers[1].bias.double())
        got = torch.nn.functional.linear(s, folded.final_layer.weight.double(), folded.final_layer.bias.double())
    # ... up to the float32 rounding of the folded weights and biases
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 3e-6 * scale, ((got - want).abs().max().item(), scale)
    with torch.no_grad():
        net.blocks[2].batch_norm_layers[0].weight[7] = -0.5
    assert ops.fold_batch_norm(net) is None
    with torch.no_grad():
        net.blocks[2].batch_norm_layers[0].weight[7] = 0.0
    assert ops.fold_batch_norm(net) is None
    ctx = ResidualNet(6, 5, hidden_features=8, context_features=3, use_batch_norm=True).eval()
    assert ops.fold_batch_norm(ctx) is None


@pytest.mark.parametrize("features,hidden,first", [(10, 32, 0), (23, 256, 0), (23, 64, 7), (9, 20, 5)])
def test_made_output_packing_reproduces_the_output_layer(features, hidden, first):
    """ops.pack_made_output (K13's weights): summing the three bf16 pieces of every packed element and walking the
    tiles the way the kernel's lanes do -- row i of a 32-row tile lands in accumulator register 4 (i // 8) + i % 4 of
    lane-half (i // 4) % 2; the 48 registers a lane-half gets from a group's three tiles are the 24 + 24 logits of
    features 4 g + 2 half and 4 g + 2 half + 1; k column = half * 128 + 8 ks + j -- gives hidden @ (W * mask)^T + b
    of the reference's MADE output layer (made.py:261-268, :282) for the features from `first` on."""
    from nflows_amd import ops
    from nflows_amd.transforms import made as made_module
    torch.manual_seed(features * 100 + hidden)
    P = 23
    net = made_module.MADE(features=features, hidden_features=hidden, num_blocks=1, output_multiplier=P)
    with torch.no_grad():
        net.final_layer.bias.normal_()
    h = torch.randn(5, hidden)
    with torch.no_grad():
        want = net.final_layer(h).view(5, features, P)[:, first:]
    wp, bp, nf = ops.pack_made_output(net, P, first_feature=first)
    assert nf == features - first
    groups = (nf + 7) // 8 * 2
    tiles = groups * 3
    assert wp.shape == (tiles, 3, 16, 2, 32, 8) and wp.dtype == torch.bfloat16 and bp.shape == (tiles, 2, 4, 4)
    w = wp.float().sum(dim=1)                                   # [tile, ks, half, r, j]
    rows = w.permute(0, 3, 2, 1, 4).reshape(tiles, 32, 256)     # [tile, r, k = half*128 + ks*8 + j]
    hp = torch.cat((h, h.new_zeros(5, 256 - hidden)), dim=1)
    i = torch.arange(32)
    reg, lane_half = 4 * (i // 8) + i % 4, (i // 4) % 2
    got = torch.zeros(5, groups * 4, 24)
    for g in range(groups):
        acc = torch.zeros(5, 3, 2, 16)
        for t in range(3):
            tile = g * 3 + t
            val = hp @ rows[tile].t()                            # [5, 32]
            acc[:, t, lane_half, reg] = val + bp[tile][lane_half, reg // 4, reg % 4]
        for half in range(2):
            got[:, 4 * g + 2 * half] = torch.cat((acc[:, 0, half], acc[:, 1, half, :8]), dim=1)
            got[:, 4 * g + 2 * half + 1] = torch.cat((acc[:, 1, half, 8:], acc[:, 2, half]), dim=1)
    scale = want.abs().max().item()
    assert (got[:, :nf, :P] - want).abs().max().item() < 2e-6 * max(scale, 1.0)
    assert got[:, :nf, P:].abs().sum().item() == 0.0 and got[:, nf:].abs().sum().item() == 0.0


def test_layer_tables_follow_the_fused_permutations():
    from nflows_amd import ops
    D = 12
    g = torch.Generator().manual_seed(4)
    perm, scat = torch.randperm(D, generator=g), torch.randperm(D, generator=g)
    tidx, iidx = torch.arange(0, D, 2), torch.arange(1, D, 2)
    t = ops.coupling_layer_tables(D, tidx, iidx, perm, scat)
    assert t.dtype == torch.int32 and t.shape == (256,)
    x = torch.randn(3, D)                         # tile slot j = input column j
    layer_in = x[:, perm]                         # what Permutation.forward hands to the layer
    assert torch.equal(x[:, t[:iidx.numel()].long()], layer_in[:, iidx])
    assert torch.equal(x[:, t[64:64 + tidx.numel()].long()], layer_in[:, tidx])
    final = torch.empty_like(x)
    final[:, scat] = layer_in                     # Permutation.inverse after a pass-through layer
    assert torch.equal(x[:, t[128:128 + D].long()], final)
    ident = ops.coupling_layer_tables(D, tidx, iidx)
    assert torch.equal(ident[128:128 + D].long(), torch.arange(D))
    # a run of three layers: emulate the data movement with plain tensors
    layers, ref = [], x.clone()
    tile = x.clone()
    perms = [torch.randperm(D, generator=g) for _ in range(3)]
    for i, p_ in enumerate(perms):
        ti, ii = (tidx, iidx) if i % 2 == 0 else (iidx, tidx)
        layers.append((ti, ii, p_, None))
    tabs = ops.flow_layer_tables(D, layers).view(4, 128).long()
    for i, (ti, ii, p_, _) in enumerate(layers):
        ref = ref[:, p_]
        assert torch.equal(tile[:, tabs[i, :ii.numel()]], ref[:, ii])
        assert torch.equal(tile[:, tabs[i, 64:64 + ti.numel()]], ref[:, ti])
        ref = ref.clone()
        ref[:, ti] = ref[:, ti] * 2 + 1            # stand-in for the spline
        tile[:, tabs[i, 64:64 + ti.numel()]] = tile[:, tabs[i, 64:64 + ti.numel()]] * 2 + 1
    assert torch.equal(tile[:, tabs[3, :D]], ref)


def test_padded_geometry_of_a_run_with_two_splits():
    """ops.fused_geometry / flow_layer_tables(padded_*) / the packers' pad_* arguments: a run over an odd feature
    count under alternating masks (layers of two splits) gets ONE geometry -- row length and transformed count in
    multiples of four, one identity count --, surplus slots of either kind all point at the first pad column (which
    no layer moves), surplus transformed features have zero rows in the packed final layer, surplus identity
    features zero columns in the initial layer (in front of the context columns when there is a context)."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    D = 21
    even, odd = torch.arange(0, D, 2), torch.arange(1, D, 2)        # 11 / 10
    assert ops.fused_geometry(64, [(32, 32), (32, 32)], 3.0) == (64, 32, 32, 7.0)          # nothing to pad
    assert ops.fused_geometry(6, [(3, 3)], 3.0) == (8, 4, 3, 7.0)                          # a spare column + rounding
    assert ops.fused_geometry(20, [(10, 10)], 1.0) == (24, 12, 10, 3.0)
    Dp, dt4, di_u, value = ops.fused_geometry(D, [(11, 10), (10, 11), (11, 10)], 3.0)
    assert (Dp, dt4, di_u, value) == (24, 12, 11, 7.0)
    g = torch.Generator().manual_seed(5)
    perms = [torch.randperm(D, generator=g) for _ in range(3)]
    layers = [((even, odd) if i % 2 == 0 else (odd, even)) + (perms[i], None) for i in range(3)]
    tabs = ops.flow_layer_tables(D, layers, padded_features=Dp, padded_transform=dt4, padded_identity=di_u).view(4, 128).long()
    x = torch.randn(5, D, generator=g)
    tile = torch.cat((x, torch.full((5, Dp - D), value)), dim=1)    # what ops._pad_columns hands to the kernel
    ref = x.clone()
    for i, (ti, ii, p_, _) in enumerate(layers):
        ref = ref[:, p_]
        assert torch.equal(tile[:, tabs[i, :ii.numel()]], ref[:, ii])
        assert torch.equal(tile[:, tabs[i, 64:64 + ti.numel()]], ref[:, ti])
        assert (tabs[i, ii.numel():di_u] == D).all() and (tabs[i, 64 + ti.numel():64 + dt4] == D).all()   # the spare column
        ref = ref.clone()
        ref[:, ti] = ref[:, ti] * 2 + 1            # stand-in for the spline (identity outside the box: the spare stays)
        tile[:, tabs[i, 64:64 + ti.numel()]] = tile[:, tabs[i, 64:64 + ti.numel()]] * 2 + 1
    assert torch.equal(tile[:, tabs[3, :D]], ref)
    assert torch.equal(tabs[3, D:Dp], torch.arange(D, Dp)) and (tile[:, D:] == value).all()   # pad columns never move
    # packers: the 10-feature layer of the run (dt = 10 -> 12, d_i = 11 is the run's count already)
    P = 23
    torch.manual_seed(0)
    net = ResidualNet(11, 10 * P, hidden_features=64, num_blocks=1)
    w_run, b_run = ops.pack_resnet_conditioner(net, 10, P, pad_transform_to=dt4, pad_identity_to=di_u)
    w_own, b_own = ops.pack_resnet_conditioner(net, 10, P, pad_transform_to=12)
    assert torch.equal(w_run, w_own) and torch.equal(b_run, b_own)
    assert w_run.shape[0] == 2 + 16 + 2 * (12 * 24 // 32) and b_run.numel() == 128 + 256 + 12 * 24
    net11 = ResidualNet(10, 11 * P, hidden_features=64, num_blocks=1)   # the other split: d_i = 10 padded to 11
    assert ops._initial_weight(net11, 11).shape == (128, 11) and (ops._initial_weight(net11, 11)[:, 10] == 0).all()
    ctx_net = ResidualNet(10, 11 * P, hidden_features=64, context_features=3, num_blocks=1)
    wi = ops._initial_weight(ctx_net, 12)           # [identity 10 | zeros 2 | context 3]
    assert wi.shape == (128, 15) and (wi[:, 10:12] == 0).all()
    assert torch.equal(wi[:64, 12:], ctx_net.initial_layer.weight.detach()[:, 10:])
    f16_w, f16_p = ops.pack_resnet_conditioner_f16(net, 10, P, pad_transform_to=dt4, pad_identity_to=di_u)
    assert f16_w.shape[0] == 1 + 8 + 12 * 24 // 32 and f16_p.numel() == 132 * 3 + 4 + 12 * 24


def test_packed_weight_caches_follow_weight_updates():
    """The re-tiled conditioner weights of the whole-layer kernels are cached per layer.  In-place
    updates under no_grad and load_state_dict are noticed; a write through `.data` is not (it does
    not advance the version counter) until `nflows_amd.invalidate_packed_weights()` is called."""
    import torch
    import nflows_amd
    from nflows_amd import configs
    flow = configs.rq_nsf_flow(num_layers=2, features=8, num_bins=8, hidden_features=128, seed=0).eval()
    layer = flow._transform._transforms[1]
    w0 = layer._packed_resnet()[0].clone()
    assert layer._packed_resnet()[0].data_ptr() == layer._packed_resnet()[0].data_ptr()  # cached
    with torch.no_grad():
        layer.transform_net.final_layer.weight.mul_(2.0)
    w1 = layer._packed_resnet()[0].clone()
    assert not torch.equal(w0, w1)
    v = layer.transform_net.final_layer.weight._version
    layer.transform_net.final_layer.weight.data.mul_(0.5)       # invisible to the version counter ...
    assert layer.transform_net.final_layer.weight._version == v
    # ... but not to the caches (round 6: the conditioner's Parameters report the use of their `.data`, the next key
    # compares contents): repacked from the new values, nothing to call
    assert torch.equal(layer._packed_resnet()[0], w0)
    # a write that announces itself through nothing at all (raw storage; here: the base class's descriptor) is still
    # served stale until nflows_amd.invalidate_packed_weights() (or the periodic checksum raises)
    torch._C.TensorBase.data.__get__(layer.transform_net.final_layer.weight).mul_(2.0)
    assert torch.equal(layer._packed_resnet()[0], w0)
    nflows_amd.invalidate_packed_weights()
    assert torch.equal(layer._packed_resnet()[0], w1)
    torch._C.TensorBase.data.__get__(layer.transform_net.final_layer.weight).mul_(0.5)
    nflows_amd.invalidate_packed_weights()
    assert torch.equal(layer._packed_resnet()[0], w0)
    sd = {k: v.clone() for k, v in flow.state_dict().items()}
    sd["_transform._transforms.1.transform_net.final_layer.weight"] *= 3.0
    flow.load_state_dict(sd)
    assert not torch.equal(layer._packed_resnet()[0], w0)


def test_fusion_planning_of_composite_transforms():
    """CompositeTransform._collect_run (the host's decision which layers go to a whole-layer kernel as one run)
    needs no device: shapes inside and outside the kernels' family, the run's padded geometry, the engine."""
    from nflows_amd import configs
    from nflows_amd.flows.base import Flow
    from nflows_amd.distributions.normal import StandardNormal
    from nflows_amd.nn.nets import MLP
    from nflows_amd.transforms import (AffineCouplingTransform, CompositeTransform, RandomPermutation,
                                       PiecewiseRationalQuadraticCouplingTransform as RQ)
    from nflows_amd.utils.torchutils import create_alternating_binary_mask

    def plan(flow, x, context=None):
        layers = list(flow._transform._transforms)
        with torch.no_grad():
            units, after = flow._transform._collect_run(layers, 0, x, context, inverse=False)
        return units, after, len(layers)

    # the headline flow: all 32 [permutation, coupling] pairs are one run, nothing to pad, the f16 engine
    flow = configs.rq_nsf_flow(num_layers=32, features=64, num_bins=8, hidden_features=128, seed=0).eval()
    units, after, n = plan(flow, torch.randn(8, 64))
    assert len(units) == 32 and after == n == 64
    first = units[0][0]
    assert first._fused_geometry(tuple(c for c, _ in units[1:])) == (64, 32, 32, 2.0 * first.tail_bound + 1.0)
    assert first._use_f16() == (RQ.conditioner_engine == "f16x2")
    # ... not with gradients, not in float64, and the inverse direction pairs the other way round
    units_grad, _ = flow._transform._collect_run(list(flow._transform._transforms), 0, torch.randn(8, 64), None, False)
    assert units_grad == []
    assert plan(flow, torch.randn(8, 64, dtype=torch.float64))[0] == []
    with torch.no_grad():
        rev = list(reversed(list(flow._transform._transforms)))
        units_inv, after_inv = flow._transform._collect_run(rev, 0, torch.randn(8, 64), None, inverse=True)
    assert len(units_inv) == 32 and after_inv == 64 and all(p is not None for _, p in units_inv)
    # odd feature count under alternating masks: two splits, one padded geometry; 10 bins; a narrow conditioner
    flow = configs.rq_nsf_flow(num_layers=4, features=21, num_bins=10, hidden_features=50, seed=1).eval()
    units, after, n = plan(flow, torch.randn(300, 21))
    assert len(units) == 4 and after == n
    assert units[0][0]._fused_geometry(tuple(c for c, _ in units[1:]))[:3] == (24, 12, 11)
    # outside the family: hidden width 256, other tails, more than 128 features
    assert plan(configs.rq_nsf_flow(num_layers=3, features=16, num_bins=8, hidden_features=256, seed=2).eval(),
                torch.randn(8, 16))[0] == []
    assert plan(configs.rq_nsf_flow(num_layers=3, features=132, num_bins=8, hidden_features=128, seed=2).eval(),
                torch.randn(8, 132))[0] == []
    # the engine of conditional layers: K8h up to 32 context features beside up to 32 identity features
    c16 = configs.conditional_rq_nsf_flow(num_layers=2, features=16, num_bins=10, hidden_features=50,
                                          raw_context=5, context_features=12, seed=3)
    c80 = configs.conditional_rq_nsf_flow(num_layers=2, features=80, num_bins=8, hidden_features=128,
                                          raw_context=5, context_features=12, seed=3)
    c40 = configs.conditional_rq_nsf_flow(num_layers=2, features=16, num_bins=8, hidden_features=128,
                                          raw_context=5, context_features=40, seed=3)
    engine = RQ.conditioner_engine == "f16x2"
    assert c16._transform._transforms[1]._use_f16() == engine
    assert not c80._transform._transforms[1]._use_f16() and not c40._transform._transforms[1]._use_f16()
    # affine layers with MLP conditioners: runs need one split (even feature counts under alternating masks)
    def affine(features):
        torch.manual_seed(features)
        layers = []
        for i in range(3):
            layers.append(RandomPermutation(features))
            layers.append(AffineCouplingTransform(create_alternating_binary_mask(features, even=(i % 2 == 0)),
                                                  lambda a, b: MLP([a], [b], [64, 64])))
        return Flow(CompositeTransform(layers), StandardNormal([features])).eval()
    assert len(plan(affine(6), torch.randn(8, 6))[0]) == 3
    assert plan(affine(7), torch.randn(8, 7))[0] == []


def test_run_plans_are_cached_and_follow_changes(monkeypatch):
    """The plan of a run (which layers, their packed weights) is kept between calls -- planning and walking the
    conditioners' parameters cost more host time than a small batch takes on the GPU -- and is re-made when
    something it depends on changes: a layer attribute, a conditioner's mode, an A/B switch, a weight (in place),
    a Parameter object replaced."""
    from nflows_amd import _cache, configs
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from nflows_amd.transforms.coupling import _weights_key
    flow = configs.rq_nsf_flow(num_layers=4, features=16, num_bins=8, hidden_features=64, seed=0).eval()
    comp = flow._transform
    x = torch.randn(8, 16)

    def plan():
        with torch.no_grad():
            return comp._collect_run(list(comp._transforms), 0, x, None, inverse=False)
    units, after = plan()
    assert len(units) == 4 and after == 8
    assert plan()[0] is units                                   # served from the cache
    assert units.geometry() == (16, 8, 8, 7.0) and units.geometry() is units.geometry()
    # a layer attribute the signature holds: the run now ends in front of that layer
    third = units[2][0]
    third.tail_bound = 5.0
    assert len(plan()[0]) == 2
    third.tail_bound = 3.0
    assert len(plan()[0]) == 4
    # a conditioner with active dropout leaves the family
    third.transform_net.blocks[0].dropout.p = 0.1
    third.transform_net.train()
    assert len(plan()[0]) == 2
    third.transform_net.eval()
    assert len(plan()[0]) == 4
    third.transform_net.blocks[0].dropout.p = 0.0
    # the A/B switches are part of the key
    monkeypatch.setattr(RQ, "fuse_conditioner", False)
    assert plan()[0] == []
    monkeypatch.setattr(RQ, "fuse_conditioner", True)
    assert len(plan()[0]) == 4
    for c, _ in units:                                          # ... and are honoured when set on the instances
        monkeypatch.setattr(c, "fuse_conditioner", False, raising=False)
    assert plan()[0] == []
    for c, _ in units:
        monkeypatch.delattr(c, "fuse_conditioner")
    assert len(plan()[0]) == 4
    # weights: version counters of a list made once per epoch; a replaced Parameter advances the epoch
    first = units[0][0]
    k0 = _weights_key(first, first.transform_net)
    assert _weights_key(first, first.transform_net) == k0
    with torch.no_grad():
        first.transform_net.final_layer.bias.add_(1.0)
    k1 = _weights_key(first, first.transform_net)
    assert k1 != k0
    bias = first.transform_net.final_layer.bias
    bias.data = bias.data.clone()                  # storage rebound, counter untouched: the pointer is in the key
    assert _weights_key(first, first.transform_net) != k1
    k1 = _weights_key(first, first.transform_net)
    epoch = _cache.epoch()
    first.transform_net.final_layer.bias = torch.nn.Parameter(torch.zeros_like(first.transform_net.final_layer.bias))
    assert _cache.epoch() > epoch
    k2 = _weights_key(first, first.transform_net)
    assert k2 != k1 and first.__dict__["_weights_list"][2][-1][2] is first.transform_net.final_layer.bias
    # swaps that bypass the registration hooks (torch.func.functional_call / stateless._reparametrize_module):
    # the held objects are checked against the modules' `_parameters` entries on every call
    from torch.nn.utils import stateless
    net = first.transform_net
    other = {n: p.detach().clone() + 1.0 for n, p in net.named_parameters()}
    epoch = _cache.epoch()
    with stateless._reparametrize_module(net, other):
        inside = _weights_key(first, net)
        assert _cache.epoch() == epoch and inside != k2
        assert inside[1:-1] == tuple((p.data_ptr(), p._version) for p in net.parameters()) and inside[-1] == 0   # (+ the `.data` salt)
    assert _weights_key(first, net) == k2
    w_before = first._packed_resnet()[1].clone()
    with torch.no_grad():
        first.transform_net.final_layer.bias.add_(2.0)
    assert not torch.equal(first._packed_resnet()[1], w_before)
    assert len(plan()[0]) == 4


def test_masked_weights_are_kept_between_no_grad_passes():
    """MaskedLinear (made.py:71-72: `F.linear(x, weight * mask, bias)`): on no-grad passes the product is formed
    once and kept until the weight changes (same rules as the packed-weight caches); with grad enabled every call
    forms it again, so autograd sees it and masked entries get zero gradients."""
    import nflows_amd
    from nflows_amd.transforms.made import MADE
    torch.manual_seed(0)
    m = MADE(features=6, hidden_features=16, num_blocks=2, output_multiplier=3).eval()
    x = torch.randn(5, 6)
    lin = m.final_layer
    with torch.no_grad():
        y0 = m(x)
        kept = lin.__dict__["_masked_weight_cache"][1]
        assert torch.equal(m(x), y0) and lin.masked_weight().data_ptr() == kept.data_ptr()
        assert torch.equal(kept, lin.mask * lin.weight)
        lin.weight.mul_(2.0)                      # in place under no_grad: the version counter advances
        y2 = m(x)
        assert not torch.equal(y2, y0) and torch.equal(lin.masked_weight(), lin.mask * lin.weight)
        lin.weight.data.mul_(0.5)                 # a write the version counter does not see
        assert torch.equal(m(x), y2)
        nflows_amd.invalidate_packed_weights()
        assert torch.equal(m(x), y0)
    y = m(x)
    assert y.requires_grad and lin.masked_weight().data_ptr() != lin.__dict__["_masked_weight_cache"][1].data_ptr()
    y.sum().backward()
    assert (lin.weight.grad[lin.mask == 0] == 0).all() and lin.weight.grad.abs().sum() > 0


def test_affine_mlp_packing_orders_the_output_rows_per_lane():
    """Host side of K11 (ops.pack_mlp_conditioner): the output layer's rows are ordered so that a
    lane-half's 16 accumulator registers of a tile are [8 shifts | the 8 scales of the same features]
    (affine) or 16 shifts (additive); padding rows are zero; stage and bias counts as the header says."""
    from nflows_amd import ops
    from nflows_amd.nn.nets import MLP
    for dt, di, additive, hidden in [(16, 16, False, (128, 128)), (20, 12, False, (128,)), (20, 44, True, (128, 128, 128))]:
        order = ops._affine_row_order(dt, additive)
        per_tile = 32 if additive else 16
        tiles = (dt + per_tile - 1) // per_tile
        assert order.shape == (tiles * 32,)
        used = order[order >= 0]
        assert sorted(used.tolist()) == list(range(dt if additive else 2 * dt))   # every output exactly once
        for t in range(tiles):
            for i in range(32):
                half, q = (i >> 2) & 1, ((i >> 3) << 2) | (i & 3)
                f = (32 * t + 16 * half + q) if additive else (16 * t + 8 * half + (q & 7))
                want = -1 if f >= dt else (f if additive or q < 8 else dt + f)
                assert order[t * 32 + i].item() == want
        torch.manual_seed(0)
        net = MLP([di], [dt if additive else 2 * dt], list(hidden))
        w, b = ops.pack_mlp_conditioner(net, dt, additive=additive)
        init_ks = 4 if di > 32 else 2
        assert w.shape == (init_ks + 8 * (len(hidden) - 1) + 2 * tiles, 768 * 8) and w.dtype == torch.bfloat16
        assert b.shape == (128 * len(hidden) + 32 * tiles,)
        # the biases of the output tiles in accumulator order: [tile][half][16]
        bo = torch.cat((net._output_layer.bias.detach(), torch.zeros(1)))
        got = b[128 * len(hidden):].view(tiles, 2, 16)
        for t in range(tiles):
            for half in range(2):
                for q in range(16):
                    i = 8 * (q // 4) + 4 * half + q % 4
                    assert got[t, half, q] == bo[order[t * 32 + i]]


def test_affine_couplings_with_residual_conditioners_route_to_k11():
    """Round 5: the reference's RealNVP composition (flows/realnvp.py:17-71: affine / additive couplings on a flipping +-1
    mask, ResidualNet conditioners).  Host side: which layers K11 takes (`_conditioner_shape` / `_run_kind`), that the
    ResidualNet is packed as the MLP made of the same Linears (initial_layer, the blocks' linear_layers in order,
    final_layer: the kernel tells the two apart by NFA_FLAG_RESIDUAL_BLOCKS), the flag's value, the planner's run."""
    import torch
    from nflows_amd import _native as N, configs, ops
    from nflows_amd.nn.nets import MLP, ResidualNet
    from nflows_amd.transforms import AdditiveCouplingTransform, AffineCouplingTransform
    header = open(os.path.join(ROOT, "include", "nflows_amd.h")).read()
    assert int(re.search(r"#define\s+NFA_FLAG_RESIDUAL_BLOCKS\s+(\d+)", header).group(1)) == N.FLAG_RESIDUAL_BLOCKS == 64
    F = torch.nn.functional
    mask = torch.ones(16)
    mask[::2] = -1

    def layer(cls=AffineCouplingTransform, **kw):
        return cls(mask, lambda i, o: ResidualNet(i, o, hidden_features=kw.pop("hidden", 128), num_blocks=2, **kw)).eval()

    with torch.no_grad():
        plain = layer()
        assert plain._conditioner_shape() == (4, True) and plain._run_kind(None) == "k11"
        assert layer(AdditiveCouplingTransform)._run_kind(None) == "k11" and layer(hidden=48)._run_kind(None) == "k11"
        assert layer(hidden=160)._run_kind(None) is None                       # wider than the kernel's 128
        assert layer(use_batch_norm=True)._run_kind(None) is None              # (not folded for K11)
        assert layer(activation=F.elu)._run_kind(None) is None                 # the kernel's blocks are ReLU
        assert layer(context_features=3)._run_kind(None) is None and plain._run_kind(torch.zeros(4, 3)) is None
        dropped = layer(dropout_probability=0.3)
        assert dropped._run_kind(None) == "k11"                                # eval mode: inactive
        dropped.train()
        assert dropped._run_kind(None) is None
        mlp = AffineCouplingTransform(mask, lambda i, o: MLP([i], [o], [128, 128])).eval()
        assert mlp._conditioner_shape() == (1, False) and mlp._run_signature() != plain._run_signature()
    assert plain._run_kind(None) is None                                       # (under autograd: the layer-by-layer path)
    # the packed stream of a ResidualNet = that of the MLP with the same Linears in the same places
    torch.manual_seed(1)
    net = ResidualNet(8, 16, hidden_features=128, num_blocks=2)
    twin = MLP([8], [16], [128] * 5)
    with torch.no_grad():
        linears = [net.initial_layer] + [lin for b in net.blocks for lin in b.linear_layers] + [net.final_layer]
        for dst, src in zip([twin._input_layer] + list(twin._hidden_layers) + [twin._output_layer], linears):
            dst.weight.copy_(src.weight)
            dst.bias.copy_(src.bias)
    (w0, b0), (w1, b1) = ops.pack_mlp_conditioner(net, 8), ops.pack_mlp_conditioner(twin, 8)
    assert torch.equal(w0, w1) and torch.equal(b0, b1) and w0.shape[0] == 2 + 8 * 4 + 2
    # the whole factory composition is one run for the planner (CPU tensors: planning only)
    flow = configs.simple_realnvp_flow(16, 128, 5, 2, seed=0).eval()
    with torch.no_grad():
        units, after = flow._transform._collect_run(list(flow._transform._transforms), 0, torch.zeros(256, 16), None, inverse=False)
    assert len(units) == 5 and after == 5 and all(p is None for _, p in units)


@pytest.mark.parametrize("residual,random_mask,features,hidden,blocks", [(True, False, 12, 20, 2), (True, False, 40, 16, 1),
                                                                       (False, True, 12, 24, 2), (False, False, 9, 32, 3)])
def test_made_schedule_reproduces_the_masked_network(residual, random_mask, features, hidden, blocks):
    """Host side of K12 (ops.pack_made_schedule): walk the per-step blocks the way the kernel does -- at step t
    the hidden units of degree t, layer by layer, written into the state vectors; then feature t's output rows
    on the hidden vector as it stands -- and compare every feature's parameters with the masked network run on
    the features before it (made.py:233-311).  Also: after the last block the hidden vector is the network's."""
    from nflows_amd import ops
    from nflows_amd.transforms import made
    torch.manual_seed(features * 7 + hidden)
    P = 5
    net = made.MADE(features=features, hidden_features=hidden, num_blocks=blocks, output_multiplier=P,
                    use_residual_blocks=residual, random_mask=random_mask).double()
    with torch.no_grad():
        for p_ in net.parameters():
            p_.copy_(torch.randn_like(p_) * 0.5)
    degrees = [net.initial_layer.degrees] + [b.degrees for b in net.blocks]
    T = int(max(int(d.max()) for d in degrees))
    blocks_f, block_at, layout = ops.pack_made_schedule(net.float(), T, P)
    net = net.double()
    n, is_res, final_src, stream_vec, num_vectors, Hp, Xp, max_block = layout[:8]
    cfg = [layout[8 + 5 * l:13 + 5 * l] for l in range(n)]
    assert blocks_f.numel() == int(block_at[-1]) * 256 and max_block % 256 == 0
    x = torch.randn(features, dtype=torch.float64)
    xs = torch.zeros(Xp, dtype=torch.float64)
    vecs = torch.zeros(num_vectors, Hp, dtype=torch.float64)
    want = net(x[None])[0].view(features, P)
    for t in range(T + 1):
        blk = blocks_f[int(block_at[t]) * 256:int(block_at[t + 1]) * 256]
        hdr = blk[:16].view(torch.int32)
        units, rows, out_rows, out_bias = (int(v) for v in hdr[:4])
        assert rows == 16 + 8 * units and blk.numel() >= 16 + 8 * 4      # (the kernel reads four entries ahead)
        in_layer_order = []
        for u in range(units):
            e = blk[16 + 8 * u:24 + 8 * u]
            bias = float(e[0])
            j, kp, src, dst, add_stream, set_stream, zero = (int(v) for v in e[1:].view(torch.int32))
            assert zero == 0 and [kp, src, dst, add_stream, set_stream] in cfg
            in_layer_order.append(cfg.index([kp, src, dst, add_stream, set_stream]))
            w = blk[rows:rows + kp].double()
            rows += kp
            v = float(w @ (xs[:kp] if src < 0 else vecs[src, :kp])) + bias
            if add_stream:
                v = float(vecs[stream_vec, j]) + v
            if set_stream:
                vecs[stream_vec, j] = v
            if dst >= 0:
                vecs[dst, j] = max(v, 0.0)
        assert in_layer_order == sorted(in_layer_order)
        if t == T:
            break
        assert rows == out_rows and out_bias == out_rows + P * Hp
        wf = blk[out_rows:out_bias].double().view(P, Hp)
        params = wf @ vecs[final_src] + blk[out_bias:out_bias + P].double()
        assert (params - want[t]).abs().max().item() < 1e-5 * (1 + want[t].abs().max().item()), t
        xs[t] = x[t]          # "feature t found"
    hidden_want = net.hidden(x[None])[0]
    assert (vecs[final_src, :hidden] - hidden_want).abs().max().item() < 1e-5 * (1 + hidden_want.abs().max().item())
    # features beyond T only read that final vector
    rest = net.final_layer(hidden_want[None])[0].view(features, P)[T:]
    assert (rest - want[T:]).abs().max().item() < 1e-9


def test_verify_weights_mode_detects_writes_through_data(monkeypatch):
    """A write through `.data` (EMA swap, dist.broadcast(p.data)) changes no version counter, no storage pointer and
    registers nothing.  Round 6: the conditioner's Parameters are re-classed (_cache.WatchedParameter) and report the use of
    their `.data`; the next key compares contents and, on a difference, changes (the `_data_salt`): the fused kernels take
    the NEW weights on the very next call, no exception -- the reference reads its parameters on every call
    (coupling.py:85).  NFA_VERIFY_WEIGHTS remains for writes that announce themselves through nothing (raw storage -- here
    the base class's descriptor): the periodic comparison raises."""
    import nflows_amd
    from nflows_amd import _cache, configs
    from nflows_amd.transforms import coupling as C
    raw = torch._C.TensorBase.data.__get__
    default_period = C.VERIFY_WEIGHTS_EVERY
    flow = configs.rq_nsf_flow(num_layers=2, features=8, num_bins=8, hidden_features=16, seed=0)
    layer = flow._transform._transforms[1]
    net = layer.transform_net
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 1)
    k0 = C._weights_key(layer, net)
    assert type(net.final_layer.bias) is _cache.WatchedParameter and isinstance(net.final_layer.bias, torch.nn.Parameter)
    assert C._weights_key(layer, net) == k0                 # unchanged weights: passes, same key
    with torch.no_grad():
        net.final_layer.bias.add_(1.0)                      # a visible update: new key, new checksum
    k1 = C._weights_key(layer, net)
    assert k1 != k0 and C._weights_key(layer, net) == k1
    net.final_layer.bias.data.mul_(0.5)                     # invisible to the counters, reported by the Parameter
    k1b = C._weights_key(layer, net)
    assert k1b != k1 and k1b[:-1] == k1[:-1] and k1b[-1] == k1[-1] + 1    # only the salt moved: repack, no exception
    assert C._weights_key(layer, net) == k1b
    assert float(net.final_layer.bias.data.abs().sum()) > 0               # a READ of `.data`: compared, nothing changes
    assert C._weights_key(layer, net) == k1b
    for p_ in net.parameters():
        p_.data.mul_(1.0)                                                  # writes that change nothing
    assert C._weights_key(layer, net) == k1b
    raw(net.final_layer.bias).mul_(0.5)                     # announced by nothing ...
    with pytest.raises(C.StalePackedWeights):
        C._weights_key(layer, net)                          # ... but not invisible to the periodic comparison
    nflows_amd.invalidate_packed_weights()
    k2 = C._weights_key(layer, net)
    assert k2 != k1b and C._weights_key(layer, net) == k2
    # a sign flip of one small entry (round 5's fp32 norms could not see it): the checksum is exact
    b = net.final_layer.bias
    with torch.no_grad():
        b.copy_(torch.linspace(-1.0, 2.0, b.numel()))
        b[3] = 1e-3
    k2 = C._weights_key(layer, net)
    raw(b)[3] = -1e-3
    with pytest.raises(C.StalePackedWeights):
        C._weights_key(layer, net)
    nflows_amd.invalidate_packed_weights()
    k2 = C._weights_key(layer, net)
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 3)       # every third use of the key only
    raw(net.final_layer.bias).mul_(2.0)
    raised_at = None
    for use in range(1, 5):
        try:
            assert C._weights_key(layer, net) == k2
        except C.StalePackedWeights:
            raised_at = use
            break
    assert raised_at is not None and raised_at <= 3, raised_at
    # the default (round 4): ON with a period of 256 uses -- an unannounced write is an exception within 256 calls, not
    # a silently wrong density for the rest of the run
    import os
    assert "NFA_VERIFY_WEIGHTS" in os.environ or default_period == 256
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 256)
    nflows_amd.invalidate_packed_weights()
    k3 = C._weights_key(layer, net)
    raw(net.final_layer.bias).add_(0.125)
    raised_at = None
    for use in range(1, 300):
        try:
            assert C._weights_key(layer, net) == k3
        except C.StalePackedWeights:
            raised_at = use
            break
    assert raised_at is not None and 1 <= raised_at <= 256, raised_at   # (the count starts at a per-layer offset)
    # diverged weights: a NaN is a bit pattern like any other -- no exception on unchanged NaN weights
    with torch.no_grad():
        net.final_layer.bias.fill_(float("nan"))
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 1)
    k4 = C._weights_key(layer, net)
    assert C._weights_key(layer, net) == k4


def test_select_columns_backward_equals_index_select():
    """autograd.SelectColumns (round 4): the identity split of a coupling layer under autograd -- forward is
    index_select, backward writes the gradient with index_copy_ into zeros (distinct columns) where torch's own backward
    is an atomic index_add_.  Same values and gradients as torch's, also when the input feeds a second consumer."""
    from nflows_amd import autograd as AG
    g = torch.Generator().manual_seed(4)
    x = torch.randn(7, 10, generator=g, dtype=torch.float64, requires_grad=True)
    cols = torch.tensor([9, 0, 4, 5])
    w = torch.randn(7, 4, generator=g, dtype=torch.float64)
    y = AG.select_columns(x, cols)
    assert torch.equal(y, x.index_select(1, cols))
    ((y * w).sum() + (x ** 2).sum()).backward()
    x2 = x.detach().clone().requires_grad_(True)
    ((x2.index_select(1, cols) * w).sum() + (x2 ** 2).sum()).backward()
    assert torch.equal(x.grad, x2.grad)
    with torch.no_grad():   # no graph: plain index_select
        assert AG.select_columns(x, cols).grad_fn is None


_ASSEMBLY = {}


def kernel_assembly(names):
    """gfx950 assembly of translation units of nflows_amd/csrc, compiled with the Makefile's flags (side by side, kept for
    the other disassembly tests of this module)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "nflows_amd", "csrc")
    makefile = open(os.path.join(csrc, "Makefile")).read().splitlines()
    no_slp = [ln for ln in makefile if ln.startswith("MFMA_SRCS")][0].split(":=")[1].split()

    def assembly(name):
        flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                 "-fhip-fp32-correctly-rounded-divide-sqrt"] + (["-fno-slp-vectorize"] if name in no_slp else [])
        return subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-I" + os.path.join(root, "include"), "-I" + csrc, "-S",
                               "--cuda-device-only", "-o", "-", os.path.join(csrc, name)],
                              capture_output=True, text=True, check=True).stdout

    todo = [n for n in names if n not in _ASSEMBLY]
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as pool:
            for n, asm in zip(todo, pool.map(assembly, todo)):
                _ASSEMBLY[n] = asm
    return [_ASSEMBLY[n] for n in names]


@pytest.mark.asm
def test_no_spill_between_a_join_and_its_exec_restore():
    """A miscompile of hipcc (ROCm 7.2) met in round 5: behind a per-lane `if` the register allocator placed the spill store
    of a value that is live ACROSS the `if` at the top of the join block, in front of the `s_or_b64 exec, exec, ...` that
    restores the lanes -- waves that skipped the `if` whole (exec = 0) never stored it and reloaded a stale slot later (K11's
    residual instances lost the log-determinants of all layers but the last in waves 2 and 3; the source now branches on a
    wave-uniform condition there; round 6: K8x's d_i > 32 instances did the same behind `if (tid < 128)` at the head of the
    layer loop -- wrong log-determinants of multi-layer runs and a garbage status word, found by tests/test_gpu_k8x.py --,
    its table prefetch is branch-free now).  Every kernel of the library: no scratch store between a label that an
    `s_cbranch_execz` jumps to and the exec restore of that block."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    makefile = open(os.path.join(root, "nflows_amd", "csrc", "Makefile")).read().splitlines()
    names = [ln for ln in makefile if ln.startswith("SRCS")][0].split(":=")[1].split()
    kernels = 0
    for name, asm in zip(names, kernel_assembly(names)):
        for k in re.split(r"\n(?=_Z\w+:\s)", asm):
            if not re.match(r"_Z\w+:", k):
                continue
            kernels += 1
            targets = set(re.findall(r"s_cbranch_execz (\.LBB\d+_\d+)", k))
            lines = k.split("\n")
            for i, line in enumerate(lines):
                m = re.match(r"(\.LBB\d+_\d+):", line)
                if not (m and m.group(1) in targets):
                    continue
                stores, j = [], i + 1
                while j < len(lines):
                    t = lines[j].strip()
                    if t.startswith("scratch_store"):
                        stores.append(t)
                    if "s_or_b64 exec, exec" in t or re.match(r"\.LBB|s_cbranch|s_branch|s_and_saveexec|s_endpgm", t):
                        break
                    j += 1
                assert not (stores and j < len(lines) and "s_or_b64 exec, exec" in lines[j]), (name, k.split(":")[0], m.group(1), stores)
    assert kernels > 400, kernels


def assert_no_read_lands_on_a_later_address(name, asm, minimum=200):
    """Round 5: the defect behind every "one wave in a few thousand is off" of rounds 3-5.  The fragment reads of the
    f16 whole-layer kernels are two ds_read_b128 in ONE asm statement sharing their address register; without an
    early-clobber on the first read's destination hipcc may give it the address register when the address dies there
    (`ds_read_b128 v[30:33], v32` / `ds_read_b128 v[66:69], v32 offset:1024`).  LDS data returns ~100 cycles after the
    issue, so the second read normally issues long before the first lands -- unless the wave is held between the two
    (an instruction-cache miss on cold code, found with a second stream keeping the device busy): then the second read
    takes its address from fragment data and a whole wave's tile of logits is off.  Inside every asm block: no LDS read's
    destination may hold the address of a later instruction of the block."""
    import re
    blocks = re.findall(r";;#ASMSTART(.*?);;#ASMEND", asm, flags=re.S)
    seen = 0
    for block in blocks:
        landed = set()
        for line in block.strip().splitlines():
            m = re.match(r"\s*ds_read\w* (v\[(\d+):(\d+)\]|v(\d+)), v(\d+)", line)
            if not m:
                continue
            seen += 1
            assert int(m.group(5)) not in landed, (name, block.strip())
            lo, hi = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
            landed |= set(range(lo, hi + 1))
    assert seen > minimum, (name, seen)


@pytest.mark.asm
def test_no_mfma_result_lands_on_its_own_operands():
    """hipcc (ROCm 7.2) renames the four-register accumulators of v_mfma_f32_16x16x32_f16 from instruction to
    instruction and, unless the operands are kept live, allocates a RESULT on the registers of the A fragment
    that instruction has just read (`v_mfma v[6:9], v[6:9], ...`): on the MI355X one wave in a few thousand then came
    out 1e-4 off, differently from launch to launch (K8s, round 3).  The kernels keep their operands live past the
    instruction; this test compiles the two f16 whole-layer kernels to assembly and looks at every MFMA."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "nflows_amd", "csrc")

    def regs(op):
        m = re.match(r"([va])\[(\d+):(\d+)\]", op)
        if m:
            return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
        m = re.match(r"([va])(\d+)$", op)
        return (m.group(1), {int(m.group(2))}) if m else ("?", set())

    # (round 4: K8h's instances live in four translation units -- the other bin counts and activations in three of their
    #  own --, compiled side by side here as in the Makefile)
    names = ("rqs_resnet_f16s.hip", "rqs_resnet_f16.hip", "rqs_resnet_f16_bins_a.hip", "rqs_resnet_f16_bins_b.hip",
             "rqs_resnet_f16_bins_c.hip", "rqs_resnet_f16_ctx_a.hip", "rqs_resnet_f16_ctx_b.hip", "rqs_resnet_f16x3.hip",
             "rqs_resnet_f16c.hip")

    listings = kernel_assembly(names)
    for name, asm in zip(names, listings):
        count = 0
        for line in asm.splitlines():
            m = re.match(r"\s+v_mfma_\w+ (\S+), (\S+), (\S+), ", line)
            if not m:
                continue
            count += 1
            (dk, d), (ak, a), (bk, b) = (regs(x.rstrip(",")) for x in m.groups())
            assert not (dk == ak and d & a) and not (dk == bk and d & b), (name, line.strip())
        assert count > 500, (name, count)
        assert_no_read_lands_on_a_later_address(name, asm, minimum=-1 if name in ("rqs_resnet_f16x3.hip", "rqs_resnet_f16c.hip") else 200)   # (K8x, K8c: no asm reads)
        # no packed fp32 arithmetic beside MFMA waves (wrong results in lanes 16-31 / 48-63 next to a
        # co-resident MFMA wave; the files are compiled with -fno-slp-vectorize, and the activations of round 4 are plain
        # C++ the compiler could have vectorised)
        assert not re.search(r"v_pk_(mul|add|fma)_f32", asm), name


def test_training_kernel_streams():
    """K14's packed streams: the forward stream is the hidden-layer prefix of K8's stream (same stages, same
    biases, bit for bit); every k-major stage group of the backward stream decodes to the TRANSPOSED weight of its
    Linear in the order the kernel consumes them (last block first: W_1^T, W_0^T), the tail to W_in^T in two-stage
    tiles with zero rows past d_i."""
    import torch
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    torch.manual_seed(5)
    for di, nb in ((32, 2), (12, 1), (48, 3)):
        net = ResidualNet(di, 8 * 23, 128, num_blocks=nb)
        blocks = [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)
                  for b in net.blocks]
        fwd, bias, bwd, none = ops.pack_resnet_hidden_train_reference(net.initial_layer.weight, net.initial_layer.bias, blocks)
        ref_w, ref_b = ops.pack_resnet_conditioner(net, 8, 23)
        n_fwd = (4 if di > 32 else 2) + 16 * nb
        assert none is None
        assert fwd.shape == (n_fwd, 6144) and torch.equal(fwd.view(torch.int16), ref_w[:n_fwd].view(torch.int16))
        assert torch.equal(bias, ref_b[:128 + 256 * nb])
        # with the final Linear appended (out = 184 -> 6 tiles, the last with 24 rows): the hidden stages are unchanged,
        # the tail decodes to W_f with zero rows behind it, the bias is in accumulator order
        fwd2, bias2, bwd2, fbias = ops.pack_resnet_hidden_train_reference(
            net.initial_layer.weight, net.initial_layer.bias, blocks, (net.final_layer.weight, net.final_layer.bias))
        out = net.final_layer.out_features
        ft = (out + 31) // 32
        assert torch.equal(fwd2[:n_fwd].view(torch.int16), fwd.view(torch.int16)) and fwd2.shape[0] == n_fwd + 2 * ft
        # (round 4) the backward stream of a call with the final Linear: the same prefix, then W_f^T as ceil(out / 16)
        # k-major stages over the columns of g_params in natural order (zero past `out`)
        kf = (out + 15) // 16
        assert torch.equal(bias2, bias) and bwd2.shape[0] == bwd.shape[0] + kf
        assert torch.equal(bwd2[:bwd.shape[0]].view(torch.int16), bwd.view(torch.int16))
        tail_t = bwd2[bwd.shape[0]:].view(kf, 4, 3, 2, 32, 8).float().sum(dim=2)    # (ks, tile, hf, i, j)
        wft = tail_t.permute(1, 3, 0, 2, 4).reshape(128, kf * 16)                    # [unit 32 t + i][k = ks*16 + hf*8 + j]
        assert torch.allclose(wft[:, :out], net.final_layer.weight.detach().t(), rtol=0, atol=1e-8) and not wft[:, out:].any()
        tailf = fwd2[n_fwd:].view(ft, 2, 3, 4, 2, 32, 8).float().sum(dim=2)    # (tile, hs, k4, hf, i, j)
        mf = tailf.permute(0, 4, 1, 2, 3, 5).reshape(ft * 32, 128)
        wf = torch.empty_like(mf)
        wf[:, ops._k8_column_order()] = mf
        assert torch.allclose(wf[:out], net.final_layer.weight.detach(), rtol=0, atol=1e-8) and not wf[out:].any()
        bpad = torch.cat((net.final_layer.bias.detach(), torch.zeros(ft * 32 - out)))
        assert torch.equal(fbias.view(ft, 2, 4, 4).permute(0, 2, 1, 3).reshape(-1), bpad)
        order_k = ops._k8_column_order()

        def decode_kmajor(stages):   # [8, 6144] -> the [128, 128] matrix whose pieces they hold
            p = stages.view(8, 4, 3, 2, 32, 8).float().sum(dim=2)          # (ks, t, hf, i, j)
            m = p.permute(1, 3, 0, 2, 4).reshape(128, 128)                  # rows t*32+i, columns (ks, hf, j)
            out = torch.empty_like(m)
            out[:, order_k] = m
            return out

        at = 0
        for b in reversed(net.blocks):
            for lin in (b.linear_layers[1], b.linear_layers[0]):
                got = decode_kmajor(bwd[at:at + 8])
                assert torch.allclose(got, lin.weight.detach().t(), rtol=0, atol=1e-7 * lin.weight.abs().max().item() + 1e-9)
                at += 8
        tiles = (di + 31) // 32
        tail = bwd[at:].view(tiles, 2, 3, 4, 2, 32, 8).float().sum(dim=2)    # (tile, hs, k4, hf, i, j)
        m = tail.permute(0, 4, 1, 2, 3, 5).reshape(tiles * 32, 128)
        wt = torch.empty_like(m)
        wt[:, order_k] = m
        assert torch.allclose(wt[:di], net.initial_layer.weight.detach().t(), rtol=0, atol=1e-8)
        assert not wt[di:].any() and bwd.shape[0] == 16 * nb + 2 * tiles
    # a narrower net (52 hidden units) is zero-padded into the same 128-wide streams
    net = ResidualNet(12, 40, 52, num_blocks=1)
    b = net.blocks[0]
    fwd, bias, bwd, fbias = ops.pack_resnet_hidden_train_reference(
        net.initial_layer.weight, net.initial_layer.bias,
        [(b.linear_layers[0].weight, b.linear_layers[0].bias, b.linear_layers[1].weight, b.linear_layers[1].bias)],
        (net.final_layer.weight, net.final_layer.bias))
    # (backward stream: 16 hidden stages + W_in^T's tile + W_f^T's ceil(40 / 16) = 3 k-major stages)
    assert fwd.shape == (2 + 16 + 4, 6144) and bwd.shape == (16 + 2 + 3, 6144) and bias.shape == (384,) and fbias.shape == (64,)
    wft = bwd[18:].view(3, 4, 3, 2, 32, 8).float().sum(dim=2).permute(1, 3, 0, 2, 4).reshape(128, 48)
    assert torch.allclose(wft[:52, :40], net.final_layer.weight.detach().t(), atol=1e-8) and not wft[52:].any() and not wft[:, 40:].any()
    w1t = decode_kmajor(bwd[:8])
    assert torch.allclose(w1t[:52, :52], b.linear_layers[1].weight.detach().t(), atol=1e-8) and not w1t[52:].any() \
        and not w1t[:, 52:].any()
    assert not bias.view(3, 4, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(3, 128)[:, 52:].any()


def test_training_function_wiring_with_emulated_kernels(monkeypatch):
    """autograd.ResidualNetHidden (K14's host side) with the three kernels and K10 replaced by tensor-operation
    emulations of their CONTRACTS (include/nflows_amd.h: what each array holds), on the CPU: the gradients it hands
    back -- order, slicing of padded widths, padding of the incoming gradient, zero columns for identity-feature
    counts that are not multiples of four, frozen parameters -- equal autograd through the eager modules."""
    import copy
    import torch
    from nflows_amd import autograd as AG
    from nflows_amd import ops
    from nflows_amd.nn.nets import ResidualNet
    book = {}

    def pad2(w, rows, cols):
        out = w.new_zeros(rows, cols)
        out[:w.shape[0], :w.shape[1]] = w
        return out

    def pad1(b, n):
        return torch.cat((b, b.new_zeros(n - b.shape[0])))

    def pack(w_in, b_in, blocks, final=None):   # the "streams": the weights zero-padded to 128 hidden units
        key = float(len(book))
        book[key] = dict(w_in=pad2(w_in.detach(), 128, w_in.shape[1]), b_in=pad1(b_in.detach(), 128),
                         blocks=[(pad2(w0.detach(), 128, 128), pad1(b0.detach(), 128), pad2(w1.detach(), 128, 128),
                                  pad1(b1.detach(), 128)) for w0, b0, w1, b1 in blocks],
                         final=None if final is None else (pad2(final[0].detach(), final[0].shape[0], 128), final[1].detach()))
        tag = torch.tensor([key], dtype=torch.float64)
        return tag, tag, tag, (tag if final is not None else None)

    def forward(x, fwd_w, fwd_b, num_blocks, final_bias=None, out_features=0):
        assert x.shape[1] % 4 == 0 and x.shape[0] % 128 == 0
        w = book[fwd_w.item()]
        h = x @ w["w_in"].t() + w["b_in"]
        saved = []
        for w0, b0, w1, b1 in w["blocks"]:
            t = torch.relu(h)
            u = torch.relu(t @ w0.t() + b0)
            saved += [t, u]
            h = h + u @ w1.t() + b1
        params = None
        if final_bias is not None:
            params = h @ w["final"][0].t() + w["final"][1]
            assert params.shape[1] == out_features
        return h, (torch.stack(saved) if saved else x.new_zeros(0, x.shape[0], 128)), params

    def backward(grad_hidden, bwd_w, saved, num_identity):
        assert grad_hidden.shape[1] == 128
        w = book[bwd_w.item()]
        nb = saved.shape[0] // 2
        grads = [None] * (2 * nb)
        gh = grad_hidden
        for k in reversed(range(nb)):
            w0, b0, w1, b1 = w["blocks"][k]
            ga = (gh @ w1) * (saved[2 * k + 1] > 0)
            gh = gh + (ga @ w0) * (saved[2 * k] > 0)
            grads[2 * k + 1], grads[2 * k] = ga, gh
        gx = gh @ w["w_in"]
        assert gx.shape[1] == num_identity
        return gx, (torch.stack(grads) if grads else grad_hidden.new_zeros(0, grad_hidden.shape[0], 128))

    fused_calls = []

    def backward_from_params(grad_params, bwd_w, saved, num_identity):   # nfa_resnet_backward_f32's contract
        w = book[bwd_w.item()]
        if grad_params.shape[1] % 4:
            return None
        g_hidden = grad_params @ w["final"][0]
        assert g_hidden.shape[1] == 128
        fused_calls.append(grad_params.shape[1])
        gx, grads = backward(g_hidden, bwd_w, saved, num_identity)
        return gx, grads, g_hidden

    monkeypatch.setattr(ops, "pack_resnet_hidden_train", pack)
    monkeypatch.setattr(ops, "resnet_hidden_forward", forward)
    monkeypatch.setattr(ops, "resnet_hidden_backward", backward)
    monkeypatch.setattr(ops, "resnet_backward", backward_from_params)
    monkeypatch.setattr(ops, "linear_wgrad", lambda x, gy, need_bias=True: (gy.t() @ x, gy.sum(0) if need_bias else None))
    # (round 4: the hidden Linears' weight gradients in one K10 launch pair when every gradient is wanted)
    batched_calls = []
    monkeypatch.setattr(ops, "linear_wgrad_batched", lambda problems, need_bias=True: (
        batched_calls.append(len(problems)), [(gy.t() @ x, gy.sum(0) if need_bias else None) for x, gy in problems])[1])
    torch.manual_seed(0)
    for di, H, nb, out, with_final, frozen in ((8, 128, 2, 40, True, ()), (3, 128, 1, 24, True, ()), (21, 52, 2, 40, True, ()),
                                               (12, 64, 0, 16, True, ()), (8, 128, 2, 40, False, ()), (6, 20, 3, 8, False, ()),
                                               (8, 128, 1, 40, True, ("initial_layer.bias", "blocks.0.linear_layers.1.weight",
                                                                      "final_layer.weight"))):
        net = ResidualNet(di, out, H, num_blocks=nb).double()
        with torch.no_grad():
            for b in net.blocks:
                b.linear_layers[1].weight.mul_(50.0)
        for name, p in net.named_parameters():
            p.requires_grad_(name not in frozen)
        x = torch.randn(256, di, dtype=torch.float64, requires_grad=True)
        weight = torch.randn(256, out if with_final else H, dtype=torch.float64)
        ref_net, ref_x = copy.deepcopy(net), x.detach().clone().requires_grad_(True)
        ((ref_net(ref_x) if with_final else ref_net.hidden(ref_x)) * weight).sum().backward()
        params = net._hidden_parameters() + ([net.final_layer.weight, net.final_layer.bias] if with_final else [])
        result = AG.ResidualNetHidden.apply(x, with_final, *params)
        assert result.shape == weight.shape
        (result * weight).sum().backward()
        assert torch.allclose(x.grad, ref_x.grad, rtol=1e-10, atol=1e-12)
        for (name, p), (_, q) in zip(net.named_parameters(), ref_net.named_parameters()):
            if name in frozen or (not with_final and name.startswith("final_layer")):
                assert p.grad is None, name
            else:
                assert p.grad.shape == p.shape and torch.allclose(p.grad, q.grad, rtol=1e-10, atol=1e-12), name
    # the batched entry served the nets whose hidden gradients were all wanted (2 nb problems each), never a frozen one
    assert batched_calls == [4, 2, 4, 4, 6], batched_calls
    # (round 4) every net with its final Linear went through the backward kernel's own final GEMM
    assert fused_calls == [40, 24, 40, 16, 40], fused_calls


def test_block_activation_routing():
    """Round 4: which conditioner activations the whole-layer kernels take (coupling._activation_ok): ReLU as before;
    F.leaky_relu / F.elu / torch.tanh at 8 or 10 bins without a context or batch norm; anything else (a lambda, GELU,
    blocks that differ) keeps the layer-by-layer path.  The code is part of the run signature (layers of one launch share
    it), K8s never takes another activation, and the flag bits round-trip."""
    import torch
    from nflows_amd import _native as N, ops
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.transforms import PiecewiseRationalQuadraticCouplingTransform as RQ
    from nflows_amd.utils import create_alternating_binary_mask
    F = torch.nn.functional

    def layer(act, K=8, ctx=None, bn=False):
        return RQ(create_alternating_binary_mask(16, even=True),
                  lambda i, o: ResidualNet(i, o, hidden_features=64, context_features=ctx, num_blocks=2, activation=act,
                                           use_batch_norm=bn),
                  num_bins=K, tails="linear", tail_bound=3.0).eval()

    assert [ops.activation_code(f) for f in (F.relu, torch.relu, F.leaky_relu, F.elu, torch.tanh, F.tanh, F.gelu, None)] == \
        [N.ACTIVATION_RELU, N.ACTIVATION_RELU, N.ACTIVATION_LEAKY_RELU, N.ACTIVATION_ELU, N.ACTIVATION_TANH, N.ACTIVATION_TANH, None, None]
    with torch.no_grad():
        for act, code in ((F.relu, 0), (F.leaky_relu, 1), (F.elu, 2), (torch.tanh, 3)):
            t = layer(act)
            assert t._block_activation() == code and t._activation_ok(None) and t._resnet_eligible(None)
            assert t._run_signature()[-1] == code
        assert layer(F.relu)._run_signature() != layer(F.elu)._run_signature()
        gelu = layer(F.gelu)
        assert gelu._block_activation() is None and not gelu._resnet_eligible(None)
        mixed = layer(F.elu)
        mixed.transform_net.blocks[1].activation = torch.tanh
        assert mixed._block_activation() is None and not mixed._resnet_eligible(None)
        assert layer(F.relu, K=4)._resnet_eligible(None) and not layer(F.elu, K=4)._resnet_eligible(None)
        assert not layer(F.elu, bn=True)._activation_ok(None) and layer(F.relu, bn=True)._activation_ok(None)
        # with a context (round 5: context instances for the other activations at 8 / 10 bins, and for every served bin
        # count with ReLU -- csrc/rqs_resnet_f16_ctx_{a,b}.hip, rqs_resnet_ctx.hip)
        ctx = torch.zeros(4, 3)
        assert layer(F.tanh, ctx=3)._activation_ok(ctx) and layer(F.relu, ctx=3)._activation_ok(ctx)
        assert not layer(F.tanh, K=4, ctx=3)._activation_ok(ctx) and layer(F.relu, K=4, ctx=3)._activation_ok(ctx)
    assert (N.ACTIVATION_TANH << N.FLAG_ACTIVATION_SHIFT) == 0x3000
    assert not ops.use_tile16(1024, 8, None, torch.device("cpu"), N.ACTIVATION_ELU)


def test_output_gradient_images_are_swizzled_conflict_free():
    """K14's backward kernel from d loss / d params (csrc/resnet_train.hip: gp_request / kstep_final): the wave's 32 rows
    x 16 columns of g_params per k-step arrive by LDS-DMA -- lane l of request i writes LDS position l of its 1-KB half
    and FETCHES chunk (l & 3) ^ ((row >> 1) & 3) of row 16 i + l / 4 -- and lane (half, r) reads chunks 2 half, 2 half + 1
    of its row at position chunk ^ ((r >> 1) & 3).  Emulated here: every reader gets exactly its eight columns
    k = 16 ks + 8 half + j (the B operand's layout, as in the forward kernel's initial layer), columns past out_features
    come from the row's last chunk (their weights are zero), and eight consecutive lanes of a 16-byte read touch eight
    different bank quads (an unswizzled image would put them on two)."""
    import numpy as np
    for out_features, ks in ((736, 0), (736, 45), (40, 2), (24, 1), (4, 0)):
        gp = np.arange(32 * out_features, dtype=np.int64).reshape(32, out_features)   # value = row * out + column
        image = np.full(32 * 16, -1, dtype=np.int64)                                  # 2 KB of floats: [row][4 chunks][4]
        for i in range(2):
            for lane in range(64):
                r = 16 * i + (lane >> 2)
                chunk = (lane & 3) ^ ((r >> 1) & 3)
                col = ks * 16 + chunk * 4
                col = col if col < out_features else out_features - 4
                dst = i * 256 + lane * 4                                               # floats: request i, LDS position = lane
                image[dst:dst + 4] = gp[r, col:col + 4]
        assert (image >= 0).all()
        for lane in range(64):
            half, r = lane >> 5, lane & 31
            sw = (r >> 1) & 3
            got = []
            for c in (2 * half, 2 * half + 1):
                pos = r * 16 + ((c ^ sw) << 2)
                got += list(image[pos:pos + 4])
            for j, v in enumerate(got):
                k = ks * 16 + 8 * half + j
                if k < out_features:
                    assert v == r * out_features + k, (out_features, ks, lane, j)
                else:   # a clamped chunk: some real value of the same row (meets a zero weight)
                    assert r * out_features <= v < (r + 1) * out_features
        # bank quads (16 bytes = four banks) of the first read of eight consecutive lanes, both halves
        for base in range(0, 64, 8):
            quads = set()
            for lane in range(base, base + 8):
                half, r = lane >> 5, lane & 31
                byte = r * 64 + (((2 * half) ^ ((r >> 1) & 3)) << 4)
                quads.add((byte // 16) % 8)
            assert len(quads) == 8, (base, sorted(quads))


def test_run_level_weight_fingerprint_and_verification(monkeypatch):
    """transforms/base.py `_Run.weights_fingerprint` (round 4): the weights of a whole run of layers in one flat pass --
    what `_run_plan` keys the run's packed weights on instead of one coupling._weights_key per layer (0.3-0.4 ms per call
    of a 32-layer flow on the host).  It follows in-place updates (version counters), rebound storage (`p.data = ...`),
    swapped parameter objects and swapped conditioners like the per-layer key; and the staggered verification (one layer
    per few calls) turns a write through `.data` into StalePackedWeights within `NFA_VERIFY_WEIGHTS` calls."""
    import torch
    import nflows_amd
    from nflows_amd import configs
    from nflows_amd.transforms import coupling as C
    flow = configs.rq_nsf_flow(4, 16, 8, 32, seed=3).eval()
    T = flow._transform
    layers = list(T._transforms)
    x = torch.zeros(128, 16)
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 8)

    def plan():
        with torch.no_grad():
            units, after = T._collect_run(layers, 0, x, None, False)
            assert after == len(layers) and len(units) == 4
            return units, T._run_plan(units, False, False)

    units, p0 = plan()
    assert plan()[1] is p0 and plan()[0] is units                      # cached: same plan object
    k0 = units.weights_fingerprint()
    net2 = layers[5].transform_net
    with torch.no_grad():
        net2.final_layer.bias.add_(1.0)                                 # in-place update: version counter
    k1 = units.weights_fingerprint()
    assert k1 != k0 and plan()[1] is not p0
    w = net2.initial_layer.weight
    w.data = w.data.clone()                                             # rebound storage, same counter
    assert units.weights_fingerprint() != k1
    k2 = units.weights_fingerprint()
    net2.initial_layer.weight = torch.nn.Parameter(w.detach().clone())  # another Parameter object (registration: epoch)
    assert plan()[0].weights_fingerprint() != k2
    p3 = plan()[1]
    assert plan()[1] is p3
    # a write through .data (round 6): the Parameter reports it, the fingerprint compares contents ONCE for the whole run
    # and the one layer whose contents changed is repacked -- a new plan on the very next call, nothing raised
    salts0 = [c.__dict__.get("_data_salt", 0) for c, _ in plan()[0]]
    layers[3].transform_net.blocks[0].linear_layers[1].weight.data.mul_(1.5)
    p3b = plan()[1]
    assert p3b is not p3 and not torch.equal(p3b[0], p3[0]) and plan()[1] is p3b
    salts = [c.__dict__.get("_data_salt", 0) for c, _ in plan()[0]]
    assert [b - a for a, b in zip(salts0, salts)] == [0, 1, 0, 0]     # (layers[3] is the second coupling: that one only)
    # a write announced by nothing (raw storage): invisible to the key, caught by the staggered verification within 8 calls
    torch._C.TensorBase.data.__get__(layers[3].transform_net.blocks[0].linear_layers[1].weight).mul_(1.5)
    with pytest.raises(C.StalePackedWeights):
        for _ in range(9):
            assert plan()[1] is p3b
    nflows_amd.invalidate_packed_weights()
    p4 = plan()[1]
    assert p4 is not p3b
    for _ in range(20):                                                 # and nothing is raised on consistent weights
        assert plan()[1] is p4


def test_user_overrides_of_the_reference_hooks_are_honoured():
    """a user subclass that overrides one of the reference's documented hooks
    (`_coupling_transform_forward / _inverse`, `_piecewise_cdf`, `_scale_and_shift`: coupling.py:132-136, :234-252,
    :279-296) -- also on the classes that have fused kernels -- must not be bypassed: such classes are marked at class
    creation (`_user_hooks`), take `_reference_sequence`, join no fused run and take no fused permutation.  (The
    sequence itself runs tensor operations on the device: the GPU suite runs it, tests/test_gpu_flows.py.)"""
    import torch
    from nflows_amd.transforms import coupling as C
    from nflows_amd.transforms.base import CompositeTransform
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.utils import create_alternating_binary_mask
    lib = (C.AffineCouplingTransform, C.AdditiveCouplingTransform, C.PiecewiseRationalQuadraticCouplingTransform,
           C.PiecewiseLinearCouplingTransform, C.PiecewiseQuadraticCouplingTransform, C.PiecewiseCubicCouplingTransform,
           C.PiecewiseCouplingTransform, C.CouplingTransform)
    mk = lambda cls, **kw: cls(create_alternating_binary_mask(6, even=True), lambda i, o: ResidualNet(i, o, 8, num_blocks=1), **kw)  # noqa: E731
    assert not any(mk(c)._user_hooks for c in lib[:2]) and not mk(lib[2], num_bins=4, tails="linear")._user_hooks

    class Plain(C.PiecewiseRationalQuadraticCouplingTransform):   # no hook touched: the fused kernels stay
        pass

    class Doubled(C.PiecewiseRationalQuadraticCouplingTransform):
        def _coupling_transform_forward(self, inputs, transform_params):
            y, lad = super()._coupling_transform_forward(inputs, transform_params)
            return 2 * y, lad

    class Child(Doubled):
        pass

    class MyScale(C.AffineCouplingTransform):
        def _scale_and_shift(self, transform_params):
            scale, shift = super()._scale_and_shift(transform_params)
            return scale + 1.0, shift

    # resolved through the MRO on every call, not at class creation -- a mixin and a function assigned
    # to the class or to the instance AFTER its creation count as well
    class HookMixin:
        def _piecewise_cdf(self, inputs, transform_params, inverse=False):
            return super()._piecewise_cdf(inputs, transform_params, inverse)

    class Mixed(HookMixin, C.PiecewiseRationalQuadraticCouplingTransform):
        pass

    class Late(C.PiecewiseRationalQuadraticCouplingTransform):
        pass

    from nflows_amd.transforms.base import _accepts_fused_permutation
    x2 = torch.zeros(4, 6)
    plain = mk(Plain, num_bins=8, tails="linear")
    assert not plain._user_hooks and _accepts_fused_permutation(plain, x2) and plain._run_kind is not None
    for cls, kw in ((Doubled, dict(num_bins=4, tails="linear")), (Child, dict(num_bins=4, tails="linear")), (MyScale, {}),
                    (Mixed, dict(num_bins=4, tails="linear"))):
        t = mk(cls, **kw)
        assert t._user_hooks and not _accepts_fused_permutation(t, x2) and not CompositeTransform._joinable(t, 6, None), cls
    late = mk(Late, num_bins=8, tails="linear")
    assert not late._user_hooks
    Late._coupling_transform_forward = Doubled._coupling_transform_forward        # assigned after class creation
    assert late._user_hooks and not CompositeTransform._joinable(late, 6, None)
    del Late._coupling_transform_forward
    assert not late._user_hooks
    late._piecewise_cdf = lambda *a, **k: None                                     # ... or on the instance
    assert late._user_hooks
    t = mk(Doubled, num_bins=4, tails="linear")
    assert not CompositeTransform._joinable(t, 6, None) and CompositeTransform._joinable(mk(Plain, num_bins=8, tails="linear"), 6, None) in (True, False)
    # the library's default hooks compute the reference's expressions (checked on the CPU: plain tensor operations)
    a = mk(MyScale)
    x, params = torch.randn(5, 3), torch.randn(5, 6)
    scale, shift = a._scale_and_shift(params)
    assert torch.allclose(scale, torch.sigmoid(params[:, 3:] + 2) + 1e-3 + 1.0) and torch.equal(shift, params[:, :3])
    y, lad = a._coupling_transform_forward(x, params)
    assert torch.allclose(y, x * scale + shift) and torch.allclose(lad, torch.log(scale).sum(1))
    xr, ladi = a._coupling_transform_inverse(y, params)
    assert torch.allclose(xr, x, atol=1e-6) and torch.allclose(ladi, -lad)
    add = mk(C.AdditiveCouplingTransform)
    s1, sh = add._scale_and_shift(torch.randn(5, 3))
    assert torch.equal(s1, torch.ones(5, 3))


def test_select_columns_backward_is_differentiable_and_checks_distinctness():
    """autograd.SelectColumns' backward is written with differentiable operations (double backward
    works), and repeated columns (a fused Permutation that is no bijection) take torch's accumulating index_select."""
    import torch
    from nflows_amd import autograd as AG
    x = torch.randn(4, 6, requires_grad=True)
    cols = torch.tensor([5, 0, 3])
    y = AG.select_columns(x, cols)
    assert torch.equal(y, x.detach()[:, cols]) and type(y.grad_fn).__name__.startswith("SelectColumns")
    (g,) = torch.autograd.grad((y ** 2).sum(), x, create_graph=True)
    want = torch.zeros(4, 6)
    want[:, cols] = 2 * x.detach()[:, cols]
    assert torch.allclose(g, want)
    (gg,) = torch.autograd.grad(g.sum(), x)           # second derivative of sum(y^2) through the custom backward
    want2 = torch.zeros(4, 6)
    want2[:, cols] = 2.0
    assert torch.allclose(gg, want2)
    dup = torch.tensor([1, 1, 2])
    yd = AG.select_columns(x, dup)
    assert not type(yd.grad_fn).__name__.startswith("SelectColumns")
    (gd,) = torch.autograd.grad(yd.sum(), x)
    assert torch.equal(gd[0], torch.tensor([0.0, 2.0, 1.0, 0.0, 0.0, 0.0]))


def test_a_plan_miss_does_not_launder_a_data_write(monkeypatch):
    """`_run_plan` used to re-record every layer's checksum on EVERY plan-cache miss -- also on
    misses that have nothing to do with the weights (the first inverse / sample() call, a batch crossing the 16-sample
    tile threshold, an evicted plan).  The per-layer packs are keyed on version counters and storage pointers, so after
    `p.data.mul_(2)` (EMA swap, dist.broadcast) the new plan was built from the OLD packs and the guard never fired: train
    -> EMA swap -> sample().  Now a miss COMPARES the layers whose key is unchanged: StalePackedWeights; after
    invalidate_packed_weights() the new plan holds the NEW weights; NaN weights are not reported as stale; a sign flip
    is seen (the checksum has a signed component)."""
    import torch
    import nflows_amd
    from nflows_amd import configs
    from nflows_amd.transforms import coupling as C
    flow = configs.rq_nsf_flow(4, 16, 8, 32, seed=3).eval()
    T = flow._transform
    layers = list(T._transforms)
    x = torch.zeros(128, 16)
    monkeypatch.setattr(C, "VERIFY_WEIGHTS_EVERY", 1 << 20)             # (the staggered check out of the way: the miss alone)

    def plan(inverse):
        with torch.no_grad():
            order = layers[::-1] if inverse else layers
            units, after = T._collect_run(order, 0, x, None, inverse)
            assert len(units) == 4
            return T._run_plan(units, inverse, False)

    raw = torch._C.TensorBase.data.__get__                              # (writes that announce themselves through nothing)
    p_fwd = plan(False)
    w = layers[3].transform_net.blocks[0].linear_layers[1].weight
    raw(w).mul_(2.0)                                                    # invisible to the keys
    # first inverse call: a new run object, whose first fingerprint compares every layer's contents with the record the
    # forward run left -- the changed layer is repacked (round 5 raised here; round 4 packed the old blobs into the new plan)
    p_inv = plan(True)
    p_fwd2 = plan(False)
    assert p_fwd2 is not p_fwd and not torch.equal(p_fwd2[0], p_fwd[0])  # packed from the doubled weights
    assert plan(True) is p_inv
    # the same write through `.data` (round 6: reported by the Parameter): the first inverse call after it packs the NEW
    # weights, nothing raised
    T.__dict__["_run_plans"].clear()
    p_fwd3 = plan(False)
    w.data.mul_(0.5)
    p_inv3 = plan(True)
    # (forward and inverse plans hold the same layers' blobs in opposite order)
    assert not torch.equal(p_inv3[0], p_inv[0]) and abs(float(p_inv3[0].float().abs().sum()) - float(plan(False)[0].float().abs().sum())) < 1e-2
    assert plan(False) is not p_fwd3
    # a sign flip leaves L1 / L2 norms alone; the exact checksum sees it
    b = layers[1].transform_net.final_layer.bias
    with torch.no_grad():
        b.copy_(torch.linspace(-1.0, 2.0, b.numel()))
    plan(False)
    T.__dict__["_run_plans"].clear()
    raw(b).neg_()
    with pytest.raises(C.StalePackedWeights):
        plan(False)
    nflows_amd.invalidate_packed_weights()
    plan(False)
    # diverged weights: NaN compares equal to NaN (the kernels propagate it like the reference), no exception
    with torch.no_grad():
        b.fill_(float("nan"))
    plan(False)
    T.__dict__["_run_plans"].clear()
    plan(False)


def test_binding_arities_match_the_header():
    """Every `nfa_*` prototype of include/nflows_amd.h against the ctypes binding: the number of arguments (a slip here
    passes garbage in registers -- no loader complains) and pointer vs integer width per position where the header is
    unambiguous (`int64_t` <-> c_int64, `int32_t` / `int` <-> c_int, pointers <-> c_void_p / POINTER)."""
    import ctypes
    import re
    from nflows_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "nflows_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(nfa_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    lib = _native.load()
    checked = 0
    for name, args in protos.items():
        fn = getattr(lib, name, None)
        if fn is None or fn.argtypes is None:
            continue
        params = [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else []
        assert len(params) == len(fn.argtypes), "%s: header %d arguments, binding %d" % (name, len(params), len(fn.argtypes))
        for pos, (decl, ct) in enumerate(zip(params, fn.argtypes)):
            pointer = "*" in decl
            if pointer:
                assert ct is ctypes.c_void_p or ct is ctypes.c_char_p or hasattr(ct, "contents") or hasattr(ct, "_type_") and \
                    isinstance(ct._type_, type), (name, pos, decl, ct)
                assert ctypes.sizeof(ct) == ctypes.sizeof(ctypes.c_void_p), (name, pos, decl, ct)
            elif "int64_t" in decl:
                assert ctypes.sizeof(ct) == 8, (name, pos, decl, ct)
            elif re.search(r"\bint32_t\b|\bint\b", decl):
                assert ctypes.sizeof(ct) == 4, (name, pos, decl, ct)
        checked += 1
    assert checked >= 40, checked


def test_piecewise_coupling_base_class_protocol():
    """coupling.py:272-296 `PiecewiseCouplingTransform`: the four spline couplings are its subclasses, and the reference's
    protocol (`_coupling_transform_forward / _inverse` reshape the conditioner output per feature, call `_piecewise_cdf`,
    row-sum) works for a user subclass that only defines `_piecewise_cdf` and `_transform_dim_multiplier`."""
    import torch
    from nflows_amd.transforms import coupling as C
    from nflows_amd.nn.nets import ResidualNet
    from nflows_amd.utils import create_alternating_binary_mask
    for cls in (C.PiecewiseRationalQuadraticCouplingTransform, C.PiecewiseLinearCouplingTransform,
                C.PiecewiseQuadraticCouplingTransform, C.PiecewiseCubicCouplingTransform):
        assert issubclass(cls, C.PiecewiseCouplingTransform) and issubclass(cls, C.CouplingTransform)
    assert C.PiecewiseRationalQuadraticCouplingTransform.supports_fused_permutation
    assert not C.PiecewiseLinearCouplingTransform.supports_fused_permutation

    class Shifted(C.PiecewiseCouplingTransform):   # y = x * exp(a) + b with per-element (a, b): a two-parameter "cdf"
        def _transform_dim_multiplier(self):
            return 2

        def _piecewise_cdf(self, inputs, transform_params, inverse=False):
            a, b = transform_params[..., 0], transform_params[..., 1]
            if inverse:
                return (inputs - b) * torch.exp(-a), -a
            return inputs * torch.exp(a) + b, a

    t = Shifted(create_alternating_binary_mask(6, even=True), lambda i, o: ResidualNet(i, o, 8, num_blocks=1))
    assert isinstance(t, C.PiecewiseCouplingTransform) and not t.supports_fused_permutation
    x = torch.randn(5, 3)
    params = torch.randn(5, 6)
    y, lad = t._coupling_transform_forward(x, params)
    p = params.reshape(5, 3, 2)
    assert torch.allclose(y, x * torch.exp(p[..., 0]) + p[..., 1]) and torch.allclose(lad, p[..., 0].sum(1))
    xr, ladi = t._coupling_transform_inverse(y, params)
    assert torch.allclose(xr, x, atol=1e-6) and torch.allclose(ladi, -lad)
    img = torch.randn(2, 3, 4, 4)
    yi, ladimg = t._coupling_transform_forward(img, torch.randn(2, 6, 4, 4))
    assert yi.shape == img.shape and ladimg.shape == (2,)
    with pytest.raises(NotImplementedError):
        C.PiecewiseCouplingTransform(create_alternating_binary_mask(4, even=True), lambda i, o: ResidualNet(i, o, 8, num_blocks=1))
