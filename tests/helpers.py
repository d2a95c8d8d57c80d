"""Shared comparison helpers for the parity tests.

Tolerance model (stated once, used everywhere)
----------------------------------------------
fp32 results of the spline depend on 1-ulp differences in exp/log through cancellation
(x - knot) and can therefore differ between two correct fp32 implementations by more than a few
ulp on ill-conditioned elements (steep bins).  The reference's own fp32 result has that error
against its fp64 result.  Parity of an fp32 implementation `got` against fp32 `ref` with fp64
ground truth `truth` is therefore asserted as:

  (a) bulk agreement: >= 99% of finite elements satisfy |got-ref| <= 2e-6 * (1 + |ref|)
      (outputs) / 2e-5 * (1 + |ref|) (log-dets);
  (b) worst case: max|got-truth| <= 4 * max|ref-truth| + 2e-6 (outputs), + 2e-5 (log-dets) -- and, since round 4,
      the MEAN and the 99.9 % QUANTILE of |got-truth| <= 2 x the reference-fp32's own (`assert_error_ratio`; floor:
      half an fp32 ulp of the typical magnitude): a maximum is set by the reference's one worst ill-conditioned
      element and hides a 10-80 x inflation of the typical element (the Newton-slope defect of round 3 passed (b));
  (c) identical NaN / inf pattern, and elements the reference passes through unchanged
      (tails) are bit-equal.
Integer / index work (permutations, untouched columns) is compared with array_equal.
"""
import os

import numpy as np

OUT_TOL = 2e-6
LAD_TOL = 2e-5


def bulk_fraction(got, ref, tol):
    fin = np.isfinite(ref)
    if fin.sum() == 0:
        return 1.0
    d = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    return float(np.mean(d <= tol * (1.0 + np.abs(ref[fin]))))


PARITY_LOG = []     # one record per assert_fp32_parity call: achieved fractions and error ratios


def _log_parity(record):
    PARITY_LOG.append(record)
    path = os.environ.get("NFA_PARITY_LOG")
    if path:
        import json
        with open(path, "a") as f:
            f.write(json.dumps(record) + "\n")


def conditioning(fn64, args, perturb, draws=3, seed=0):
    """Per-element conditioning scale of a float64 evaluation `fn64(*args)`: the largest change of every
    result element when the arguments listed in `perturb` (indices into `args`) are moved by half an fp32
    ulp in random directions (x -> x (1 + s 2^-24), s = +-1 per element), over `draws` draws.  This is
    the error an fp32 implementation inherits from rounding its INPUTS alone; roundings inside the
    evaluation add terms of the same size, so parity allows a small multiple of it."""
    rng = np.random.RandomState(seed)
    args = [np.asarray(a_, dtype=np.float64) if isinstance(a_, np.ndarray) and a_.dtype.kind == "f" else a_ for a_ in args]
    base = fn64(*args)
    base = base if isinstance(base, tuple) else (base,)
    worst = [np.zeros_like(np.asarray(b, dtype=np.float64)) for b in base]
    for _ in range(draws):
        moved = list(args)
        for i in perturb:
            a = np.asarray(args[i], dtype=np.float64)
            moved[i] = a * (1.0 + (rng.randint(0, 2, size=a.shape) * 2 - 1) * 2.0 ** -24)
        out = fn64(*moved)
        out = out if isinstance(out, tuple) else (out,)
        for w, b, o in zip(worst, base, out):
            with np.errstate(invalid="ignore"):
                d = np.abs(np.asarray(o, dtype=np.float64) - np.asarray(b, dtype=np.float64))
            np.maximum(w, np.where(np.isfinite(d), d, 0.0), out=w)
    return worst if len(worst) > 1 else worst[0]


def error_stats(err):
    err = np.asarray(err, dtype=np.float64).reshape(-1)
    return {"max": float(err.max()), "mean": float(err.mean()), "q999": float(np.quantile(err, 0.999))}


def assert_error_ratio(got, ref, truth, what="", factor=2.0, max_factor=4.0, max_floor=0.0):
    """err(got vs float64) <= `factor` x err(reference fp32 vs float64) on the mean and the 99.9 % quantile,
    `max_factor` x (+ `max_floor`) on the maximum.  Floor on mean / quantile: half an fp32 ulp of the mean magnitude
    (an fp32 result cannot be asked to sit closer to the truth than its own rounding; it matters only for vectors on
    which the reference's fp32 happens to be exact).  Returns the figures."""
    got, ref, truth = (np.asarray(a) for a in (got, ref, truth))
    fin = np.isfinite(ref) & np.isfinite(truth) & np.isfinite(got)
    if not fin.any():
        return None
    t64 = truth[fin].astype(np.float64)
    e_got = error_stats(np.abs(got[fin].astype(np.float64) - t64))
    e_ref = error_stats(np.abs(ref[fin].astype(np.float64) - t64))
    floor = 2.0 ** -24 * float(np.abs(t64).mean())
    for k in ("mean", "q999"):
        # (below 10 000 elements the 99.9 % quantile IS one of the ten worst elements: it gets the maximum's factor)
        f = max_factor if (k == "q999" and t64.size < 10000) else factor
        assert e_got[k] <= f * e_ref[k] + floor, (
            "%s: %s error vs float64 %.3e exceeds %.1f x the reference fp32's %.3e" % (what, k, e_got[k], f, e_ref[k]))
    assert e_got["max"] <= max_factor * e_ref["max"] + max_floor + floor, (
        "%s: max error vs float64 %.3e exceeds %.1f x the reference fp32's %.3e" % (what, e_got["max"], max_factor, e_ref["max"]))
    return {"got": e_got, "reference": e_ref}


def assert_fp32_parity(got, ref, truth, tol, what="", bulk=0.999, factor=4.0, cond=None, cond_factor=32.0):
    """Parity of an fp32 result `got` with the reference's fp32 `ref`, given the float64 truth.
    Always: identical NaN / inf pattern, and the worst case max |got - truth| <= factor * max |ref - truth|
    + tol (1 + max |truth|).
    With `cond` (per-element conditioning scale from `conditioning`) every element has its own allowance
        A_i = tol (1 + |truth_i|) + cond_factor * cond_i + 2 |ref_i - truth_i|
    (rounding of the inputs amplified by the element's conditioning, plus what the reference's own fp32
    evaluation loses at that element: its formulas have unstable spots -- the inverse's quadratic root
    near a knot loses 100 x more than the conditioning explains -- and an implementation of the same
    formulas shares them).  Required: >= `bulk` of the elements |got - truth| <= A_i and >= `bulk`
    |got - ref| <= 2 A_i.  Without `cond` (vectors whose generating function is not at hand): >= `bulk`
    of the elements within tol (1 + |ref|) of `ref`."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    truth = np.asarray(truth)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern differs"
    inf = np.isinf(ref)
    assert np.array_equal(got[inf], ref[inf]), what + ": inf pattern differs"
    frac = bulk_fraction(got, ref, tol)
    fin = np.isfinite(ref) & np.isfinite(truth)
    record = {"what": what, "elements": int(fin.sum()), "tol": tol, "within_tol_of_reference": frac}
    if not fin.any():
        _log_parity(record)
        return
    g64, r64, t64 = got[fin].astype(np.float64), ref[fin].astype(np.float64), truth[fin].astype(np.float64)
    e_got, e_ref = np.abs(g64 - t64), np.abs(r64 - t64)
    record.update(max_err_vs_fp64=float(e_got.max()), reference_max_err_vs_fp64=float(e_ref.max()),
                  mean_err_vs_fp64=float(e_got.mean()), reference_mean_err_vs_fp64=float(e_ref.mean()))
    worst_ok = e_got.max() <= factor * e_ref.max() + tol * (1 + np.abs(t64).max())
    if cond is None:
        _log_parity(record)
        assert frac >= bulk, "%s: only %.5f of elements within %g" % (what, frac, tol)
    else:
        c = np.asarray(cond, dtype=np.float64)[fin]
        allow = tol * (1.0 + np.abs(t64)) + cond_factor * c + 2.0 * e_ref
        frac_truth = float(np.mean(e_got <= allow))
        frac_ref = float(np.mean(np.abs(g64 - r64) <= 2.0 * allow))
        record.update(within_allowance_of_fp64=frac_truth, within_twice_allowance_of_reference=frac_ref,
                      worst_error_over_allowance=float((e_got / allow).max()))
        _log_parity(record)
        assert frac_truth >= bulk, "%s: only %.5f of elements within their allowance of the fp64 result" % (what, frac_truth)
        assert frac_ref >= bulk, "%s: only %.5f of elements within twice their allowance of the reference" % (what, frac_ref)
    assert worst_ok, "%s: max err vs fp64 %.3e, reference fp32's own %.3e" % (what, e_got.max(), e_ref.max())
    # mean and 99.9 % quantile at the 2 x rule (round 4); the maximum stays the rule above
    assert_error_ratio(got, ref, truth, what, factor=2.0, max_factor=factor, max_floor=tol * (1 + np.abs(t64).max()))


def eager_oracle(flow_cpu, x, noise=None, context=None, fp64_device=None):
    """The two evaluations every whole-flow parity test needs, both through oracle/eager.py (the op-for-op PyTorch port of
    the reference's path; tests/test_oracle_golden.py pins it bit for bit to the real reference on the CPU):
      * float32 ON THE CPU -- the reference's own arithmetic, the yardstick ("the reference-fp32's own error");
      * float64 -- the truth.  With `fp64_device` the same port is run by stock PyTorch on that device (aten's float64
        kernels, nothing of libnflows_amd): at 65 536 rows the float64 pass on the GPU box's 16 cores was 2/3 of the
        GPU suite's running time (round 5), and as a truth to 1e-12 it does not matter which of aten's float64 kernels
        rounded it (tests/test_gpu_flows.py::test_device_float64_port_is_the_reference_float64 holds the device
        evaluation to the real reference's float64 vectors).
    Returns {z32, lad32, lp32, z64, lad64, lp64} and, with `noise`, {xi32, ladi32, xi64, ladi64} (the inverse) as numpy
    arrays.  `context`: raw context rows of a conditional flow (embedded by the flow's own embedding net)."""
    import copy
    import torch
    from oracle import eager
    threads = torch.get_num_threads()
    out = {}

    def run(f, xx, nn_, ctx, tag):
        emb = None if ctx is None else f._embedding_net(ctx)
        z, lad = eager.flow_transform(f, xx, context=emb)
        lp = eager.standard_normal_log_prob(z) + lad
        res = {"z": z, "lad": lad, "lp": lp}
        if nn_ is not None:
            res["xi"], res["ladi"] = eager.flow_transform(f, nn_, inverse=True, context=emb)
        for k, v in res.items():
            out[k + tag] = v.cpu().numpy()

    with torch.no_grad():
        run(flow_cpu.float(), x.float(), None if noise is None else noise.float(), None if context is None else context.float(), "32")
        if fp64_device is None:
            f64 = flow_cpu.double()
            run(f64, x.double(), None if noise is None else noise.double(), None if context is None else context.double(), "64")
            flow_cpu.float()
        else:
            f64 = copy.deepcopy(flow_cpu).double().to(fp64_device)
            to = lambda t: None if t is None else t.double().to(fp64_device)   # noqa: E731
            run(f64, to(x), to(noise), to(context), "64")
            del f64
    torch.set_num_threads(threads)
    return out


def knot_case_keep(got_y, ref_y, got_lad, ref_lad, status, inverse, what=""):
    """The adversarial cases of rqs_bins.npz put inputs ON a knot.  In the inverse direction the discriminant
    (rational_quadratic.py:141) is then a difference of nearly equal terms and can round below zero in one correct fp32
    evaluation and not in another (the reference's own does for 3 of the generated inputs: its assertion :142 rejected
    them, make_golden.py moved them; an implementation reports it as NFA_STATUS_NEG_DISCRIMINANT and NaN), and a root one
    ulp outside [0, 1] can make the argument of a logarithm negative (the reference's own logabsdet is NaN at 6 of the
    4096 steep inputs).  Returns the mask of the elements to compare: all of them in the forward direction (no status,
    no NaN allowed), in the inverse those finite on both sides -- at least 99.5 % of them."""
    got_y, ref_y, got_lad, ref_lad = (np.asarray(a) for a in (got_y, ref_y, got_lad, ref_lad))
    if not inverse:
        assert status == 0, (what, status)
        assert np.array_equal(np.isnan(got_y), np.isnan(ref_y)) and np.array_equal(np.isnan(got_lad), np.isnan(ref_lad)), what
        return np.ones(got_y.shape, dtype=bool)
    assert status in (0, 2), (what, status)
    keep = ~(np.isnan(got_y) | np.isnan(ref_y) | np.isnan(got_lad) | np.isnan(ref_lad))
    assert keep.mean() >= 0.995, (what, float(keep.mean()))
    return keep


def assert_bins_match(got, ref, x, knots, what="", max_fraction=4e-6):
    """Bin indices of a fused search against the oracle's on random inputs.  Equal -- except that two correct fp32
    evaluations disagree on an input that sits between their two versions of a knot.  A knot is
    RN(span * cumsum + left) (rational_quadratic.py:95): 1-ulp differences of exp / log move the prefix sum by an ulp
    or two and the product with the span is rounded at magnitude ~ span, so two versions of a knot lie within a few
    ulp(span) of each other (the oracle's against the real reference's: up to 4, tests/test_oracle_golden.py) and the probability of an input in between is ~ K x 2^-22 per element.  So: at most
    `max_fraction` of the elements differ (4e-6: four per 2^20), each by ONE bin, each with the input within
    6 ulp(span) of the oracle's knot between the two bins, and the tail decision (-1: a compare with +-B) never.
    `knots` [n, K + 1]: the oracle's knots of the searched axis.  Returns the indices of the differing elements."""
    got, ref = np.asarray(got).reshape(-1).astype(np.int64), np.asarray(ref).reshape(-1).astype(np.int64)
    x = np.asarray(x).reshape(-1)
    d = np.nonzero(got != ref)[0]
    if d.size == 0:
        return d
    assert d.size <= max(1, int(np.ceil(max_fraction * got.size))), "%s: %d of %d bins differ" % (what, d.size, got.size)
    assert np.all(np.abs(got[d] - ref[d]) == 1), what
    assert not np.any((got[d] == -1) | (ref[d] == -1)), what
    knots = np.asarray(knots).reshape(got.size, -1)
    kn = knots[d, np.maximum(got[d], ref[d])]
    span = (knots[d, -1] - knots[d, 0]).astype(np.float32)
    assert np.all(np.abs(x[d].astype(np.float64) - kn.astype(np.float64)) <= 6 * np.spacing(span)), what
    return d


def parse_kwargs(text):
    import ast
    return dict(ast.literal_eval(text))


def assert_sibling_spline_parity(got, ref, truth, tol, cap, what=""):
    """Linear / quadratic splines: same bulk criterion as assert_fp32_parity (>= 97 % of the elements
    within tol * (1 + |ref|) of the reference's fp32 output), identical NaN pattern, and the error
    against the reference's float64 result bounded by max(4 x the reference-fp32's own, cap).
    `cap` is there because these splines have elements whose result moves by 1e3 ulp when one
    intermediate prefix sum moves by one ulp (a bin of minimal width next to a steep one: the
    quadratic inverse divides a difference of two nearly equal cdf values by a tiny 2a); the
    reference's own fp32 error at such an element is a matter of luck, not a bound."""
    got, ref, truth = np.asarray(got), np.asarray(ref), np.asarray(truth)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern differs"
    frac = bulk_fraction(got, ref, tol)
    assert frac >= 0.97, "%s: only %.4f of elements within %g" % (what, frac, tol)
    fin = np.isfinite(ref) & np.isfinite(truth)
    if fin.any():
        e_got = np.abs(got[fin].astype(np.float64) - truth[fin]).max()
        e_ref = np.abs(ref[fin].astype(np.float64) - truth[fin]).max()
        assert e_got <= max(4.0 * e_ref, cap), "%s: max err vs fp64 %.3e (reference fp32: %.3e)" % (what, e_got, e_ref)


CONTEXT_MORE_CASES = ("ctx_k4", "ctx_k6", "ctx_k9", "ctx_k12", "ctx_k16", "ctx_k24", "ctx_leaky_relu_k8", "ctx_elu_k10",
                      "ctx_tanh_k8", "ctx_tanh_k10")


def golden_conditional_flow(golden_dir, case=None):
    """The conditional flow of tests/golden/flows_context.npz -- or `case` of flows_context_more.npz (round 5: other bin
    counts and block activations) -- rebuilt from its seed (weights are not stored; per-parameter checksums are, and are
    checked here).  Returns (flow on CPU, npz, name)."""
    import torch
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, "flows_context.npz" if case is None else "flows_context_more.npz"))
    meta = dict((str(n), str(c)) for n, c in g["meta"])
    name = str(g["meta"][0][0]) if case is None else case
    cfg = parse_kwargs(meta[name])
    F = torch.nn.functional
    act = {"relu": F.relu, "leaky_relu": F.leaky_relu, "elu": F.elu, "tanh": torch.tanh}[cfg.get("activation", "relu")]
    flow = configs.conditional_rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], cfg["C"], cfg["E"],
                                           cfg["tail_bound"], seed=cfg["seed"], activation=act)
    with torch.no_grad():
        for n_, p in flow.named_parameters():
            if "final_layer" in n_:
                p.mul_(cfg["scale_final"])
            elif "linear_layers.1" in n_:
                p.mul_(cfg["scale_linear1"])
            elif "context_layer" in n_:
                p.mul_(cfg["scale_context"])
    sd = flow.state_dict()
    for n_, (total, absolute) in zip(g[name + "/param_names"], g[name + "/param_checksums"]):
        v = sd[str(n_)].double()
        assert abs(float(v.sum()) - total) <= 1e-9 * (1 + abs(total)), n_
        assert abs(float(v.abs().sum()) - absolute) <= 1e-9 * (1 + absolute), n_
    return flow.eval(), g, name


REALNVP_CASES = ("realnvp_affine", "realnvp_additive", "realnvp_h64_d22", "realnvp_d64_b3", "realnvp_d80")


def golden_realnvp_flow(golden_dir, case):
    """`case` of tests/golden/flows_realnvp.npz -- the reference's SimpleRealNVP factory (flows/realnvp.py:17-71) at K11's
    conditioner width -- rebuilt by configs.simple_realnvp_flow from its seed (weights are not stored; per-parameter
    checksums of the factory's weights are, and are checked here).  Returns (flow on CPU, npz, cfg)."""
    import torch
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, "flows_realnvp.npz"))
    cfg = parse_kwargs(dict((str(n), str(c)) for n, c in g["meta"])[case])
    flow = configs.simple_realnvp_flow(cfg["features"], cfg["hidden_features"], cfg["num_layers"], cfg["num_blocks_per_layer"],
                                       cfg["use_volume_preserving"], seed=cfg["seed"])
    with torch.no_grad():
        for n_, p in flow.named_parameters():
            if "final_layer" in n_:
                p.mul_(cfg["scale_final"])
            elif "linear_layers.1" in n_:
                p.mul_(cfg["scale_linear1"])
    sd = flow.state_dict()
    assert [str(n_) for n_ in g[case + "/param_names"]] == list(sd.keys()), "state_dict keys differ from the reference's"
    for n_, (total, absolute) in zip(g[case + "/param_names"], g[case + "/param_checksums"]):
        v = sd[str(n_)].double()
        assert abs(float(v.sum()) - total) <= 1e-9 * (1 + abs(total)), n_
        assert abs(float(v.abs().sum()) - absolute) <= 1e-9 * (1 + absolute), n_
    return flow.eval(), g, cfg


def steepen(module, num_bins=None, wh_scale=1.0, d_scale=1.0, hidden_scale=1.0):
    """Turns a freshly initialised (near-identity) flow into one with STEEP splines, as after training: every
    conditioner's output layer (`final_layer`, rows per transformed feature [w_0..w_{K-1}, h_0..h_{K-1}, d_1..d_{K-1}],
    coupling.py:289, 550-552) gets its width / height rows multiplied by `wh_scale` and its derivative rows by
    `d_scale`; `num_bins=None`: an affine coupling's [shift | scale logit] halves (coupling.py:235-236), the second
    half by `d_scale`.  `hidden_scale` multiplies the second Linear of every residual block (non-trivial hidden
    activations).  Works on the reference's modules and on the drop-in classes alike (same parameter names):
    tests/golden/make_golden.py and the tests call this one function."""
    import torch
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "linear_layers.1" in name:
                p.mul_(hidden_scale)
            elif "final_layer" in name or "_output_layer" in name:   # (ResidualNet / MADE; MLP: mlp.py:45)
                if num_bins is None:
                    p[p.shape[0] // 2:].mul_(d_scale)
                else:
                    P = 3 * num_bins - 1
                    assert p.shape[0] % P == 0, (name, tuple(p.shape))
                    v = p.view(p.shape[0] // P, P, *p.shape[1:])
                    v[:, :2 * num_bins].mul_(wh_scale)
                    v[:, 2 * num_bins:].mul_(d_scale)
    return module


def steep_flow(golden_dir, name, fixture="flows_steep.npz"):
    """A flow of tests/golden/flows_steep.npz (or `fixture`: flows_bins.npz, the other bin counts) rebuilt from its seed and `steepen` (weights are not stored; the
    per-parameter checksums of the reference's weights are, and are checked here).  Returns (flow on CPU, npz, cfg)."""
    import torch
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, fixture))
    cfg = parse_kwargs(dict((str(n), str(c)) for n, c in g["meta"])[name])
    if cfg["kind"] == "rq_nsf":
        F = torch.nn.functional
        act = {"relu": F.relu, "leaky_relu": F.leaky_relu, "elu": F.elu, "tanh": torch.tanh}[cfg.get("activation", "relu")]
        flow = configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"], seed=cfg["seed"], activation=act)
        steepen(flow, cfg["K"], cfg["wh_scale"], cfg["d_scale"], cfg["hidden_scale"])
    elif cfg["kind"] == "affine":
        flow = configs.affine_coupling_flow(cfg["L"], cfg["D"], tuple(cfg["hidden"]), seed=cfg["seed"])
        steepen(flow, None, d_scale=cfg["d_scale"])
    elif cfg["kind"] == "ar_rq":
        flow = configs.ar_rq_flow(cfg["D"], cfg["H"], cfg["K"], cfg["tail_bound"], cfg["num_blocks"], seed=cfg["seed"])
        steepen(flow, cfg["K"], cfg["wh_scale"], cfg["d_scale"])
    else:
        raise KeyError(cfg["kind"])
    # (the affine fixture's conditioner is the reference's MLP behind an (inputs, context) wrapper: other parameter
    #  NAMES, the same values in the same order)
    sums = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in flow.state_dict().values()])
    want = g[name + "/param_checksums"]
    assert sums.shape == want.shape and np.allclose(sums, want, rtol=1e-12, atol=0), "seeded weights differ from the reference's: " + name
    return flow.eval(), g, cfg


def trained_flow(golden_dir, name="trained_nsf"):
    """The flow of tests/golden/flows_trained.npz -- TRAINED with the reference (make_golden.py `trained`) -- as drop-in
    classes with the stored state_dict loaded strictly (same keys as the reference's: SURVEY appendix A11).  Returns
    (flow on CPU in eval mode, npz, cfg)."""
    import torch
    from nflows_amd import configs
    g = np.load(os.path.join(golden_dir, "flows_trained.npz"))
    cfg = parse_kwargs(dict((str(n), str(c)) for n, c in g["meta"])[name])
    flow = configs.rq_nsf_flow(cfg["L"], cfg["D"], cfg["K"], cfg["H"], 2, cfg["tail_bound"], seed=cfg["seed"])
    prefix = name + "/sd/"
    flow.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}, strict=True)
    return flow.eval(), g, cfg
