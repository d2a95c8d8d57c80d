"""The Transform API -- the drop-in boundary of this package.

Mirrors nflows/transforms/base.py: `Transform.forward/inverse(inputs, context=None) ->
(outputs, logabsdet)` (:22-29), `CompositeTransform` (:32-60) and `InverseTransform` (:215-231).

MI355X-specific addition: `CompositeTransform` recognises a column `Permutation` that is
adjacent to a coupling layer and hands the permutation to the coupling kernel (gather on the way
in for `forward`, scatter on the way out for `inverse`), which removes one full read+write pass
over the [batch, features] activations per layer.  Folding a permutation is bit-identical to
running the two transforms one after the other (it only changes which column a kernel reads or
writes).  The whole-layer kernel is a different matter: it computes the conditioner's GEMMs in its
own summation order, so its results agree with the layer-by-layer path to fp32 rounding, not bit
for bit.  It works on full 128-row blocks; a ragged batch is padded with zero rows whose results are
dropped, so every row takes the same kernel and its result is bit-identical whatever the batch size
(rows are independent inside a block).  `CompositeTransform.fuse_layer_runs = False` and
`PiecewiseRationalQuadraticCouplingTransform.fuse_conditioner = False` select the layer-by-layer
path everywhere.
"""
import torch
from torch import nn

from .. import _cache
from ..errors import InputOutsideDomain, InverseNotAvailable  # noqa: F401  (re-exported)


class Transform(nn.Module):
    """Base class of all transforms."""

    # weights arriving through load_state_dict / .to() / .cuda() / .double(): drop the packed copies
    def _load_from_state_dict(self, *args, **kwargs):
        _cache.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, *args, **kwargs):
        _cache.invalidate()
        return super()._apply(*args, **kwargs)

    def forward(self, inputs, context=None):
        raise NotImplementedError()

    def inverse(self, inputs, context=None):
        raise InverseNotAvailable()


def _is_column_permutation(t):
    from .permutations import Permutation
    return isinstance(t, Permutation) and t._dim == 1


def _accepts_fused_permutation(t, inputs):
    return (getattr(t, "supports_fused_permutation", False) and not getattr(t, "_user_hooks", False) and inputs.dim() == 2
            and inputs.dtype == torch.float32)   # (float64 flows take the generic device path; a user's hooks: the reference's sequence)


def _switch_state():
    """The class-level A/B switches a run plan depends on."""
    from .coupling import (AdditiveCouplingTransform, AffineCouplingTransform,
                           PiecewiseRationalQuadraticCouplingTransform as RQ)
    return (RQ.fuse_conditioner, RQ.fuse_final_linear, RQ.conditioner_engine, RQ.conditioner_act_scale,
            RQ.resnet_log2e, AffineCouplingTransform.fuse_conditioner, AdditiveCouplingTransform.fuse_conditioner)


def _layer_state(units):
    """Per-layer attributes a run plan depends on, re-read on every call: the layers' signatures (bins, tails,
    box, minimum sizes, engine ...), whether their conditioners are in training mode (active dropout), their
    unconditional transforms, and the A/B switches once more (they may be set on an instance)."""
    # (`_modules[...]`: the plain dict behind `c.transform_net`, without nn.Module's attribute fallback)
    return [(c._run_signature(), c._modules["transform_net"].training, c._modules.get("unconditional_transform") is None,
             c.fuse_conditioner, getattr(c, "fuse_final_linear", None)) for c, _ in units]


class _Run(list):
    """The units [(coupling, permutation or None)] of a run, with what is the same on every call kept beside
    them: the run's padded geometry (ops.fused_geometry over its layers)."""

    def geometry(self):
        held = self.__dict__.get("_geometry")
        if held is None:
            held = self[0][0]._fused_geometry(tuple(c for c, _ in self[1:]))
            self.__dict__["_geometry"] = held
        return held

    # ---- the weights of the whole run in one flat pass (round 4).  coupling._weights_key layer by layer cost ~10 us per
    #      layer and call -- 0.3-0.4 ms per `log_prob` of a 32-layer flow, more than the kernel takes on 8 192 rows: the
    #      small-batch figures of bench.py were the HOST's.  Same signs as _weights_key (epoch, identity of every held
    #      object in its module's dict, version counters, storage pointers), read from lists made once per epoch.
    def _flatten(self, epoch):
        from .coupling import _held_parameters
        nets = [c._modules["transform_net"] for c, _ in self]
        per_layer = [_held_parameters(net) for net in nets]
        entries = [e for held in per_layer for e in held]
        bounds, at = [], 0
        for held in per_layer:
            bounds.append((at, at + len(held)))
            at += len(held)
        flat = (epoch, nets, entries, [p for _, _, p in entries], bounds)
        self.__dict__["_flat"] = flat
        return flat

    def weights_fingerprint(self):
        from .. import _cache
        from . import coupling
        epoch = _cache.epoch()
        flat = self.__dict__.get("_flat")
        if flat is None or flat[0] != epoch or not _cache.HOOKED:
            flat = self._flatten(epoch)
        else:
            for (c, _), net in zip(self, flat[1]):
                if c._modules["transform_net"] is not net:
                    flat = self._flatten(epoch)
                    break
            else:
                for d, name, p in flat[2]:
                    if d.get(name) is not p:
                        flat = self._flatten(epoch)
                        break
        params = flat[3]
        key = (epoch, tuple([p._version for p in params]), tuple([p.data_ptr() for p in params]))
        every = coupling.VERIFY_WEIGHTS_EVERY
        de = _cache.data_epoch()
        if self.__dict__.get("_data_seen") != de:
            # the `.data` of a watched parameter was touched since the run last looked (or this is its first use): every
            # layer's record is brought up to date -- ONE checksum over all parameters of the run, one synchronising
            # comparison; layers whose contents changed under an unchanged visible key get a new `_data_salt`
            if not coupling._capturing(params):
                self.__dict__["_data_seen"] = de
                self._recheck_run(flat, key, de)
        elif every and params and not coupling._capturing(params):
            # the periodic check: one layer at a time, every layer once per `every` calls (no call pays for all of them)
            n = self.__dict__["_verify_calls"] = self.__dict__.get("_verify_calls", 0) + 1
            stride = max(1, every // len(self))
            if n % stride == 0:
                i = (n // stride) % len(self)
                lo, hi = flat[4][i]
                coupling._track_contents(self[i][0], "_contents_run", (epoch, key[1][lo:hi], key[2][lo:hi]), params[lo:hi],
                                         periodic=False, compare_now=True)
        return key + (tuple([c.__dict__.get("_data_salt", 0) for c, _ in self]),)

    def _recheck_run(self, flat, key, de):
        """A `.data` event (or the run's first use): per layer, contents on record under the layer's current visible key are
        compared with the contents now -- a difference, or no such record (the key moved since: nothing to compare with),
        advances the layer's `_data_salt`, i.e. its packs are rebuilt from the current values --, and the records are renewed."""
        from . import coupling
        params, bounds = flat[3], flat[4]
        now = coupling._checksum(params)
        verdicts, rows = [], []
        for i, ((c, _), (lo, hi)) in enumerate(zip(self, bounds)):
            sub = (flat[0], key[1][lo:hi], key[2][lo:hi])
            state = c.__dict__.get("_contents_run")
            if state is not None and state[0] == sub and state[1].shape[0] == hi - lo:
                verdicts.append((state[1] == now[lo:hi]).all())
                rows.append(i)
            else:
                c.__dict__["_data_salt"] = c.__dict__.get("_data_salt", 0) + 1
            c.__dict__["_contents_run"] = [sub, now[lo:hi], 0, de]
            layer_record = c.__dict__.get("_contents")      # (the layer's own record, coupling._weights_key: same event,
            if layer_record is not None:                     #  same contents -- it must not answer it a second time)
                layer_record[1], layer_record[3] = now[lo:hi], de
        if rows:
            for same, i in zip(torch.stack(verdicts).cpu().tolist(), rows):    # (the one synchronisation)
                if not same:
                    c = self[i][0]
                    c.__dict__["_data_salt"] = c.__dict__.get("_data_salt", 0) + 1

    def verify_before_packing(self):
        """Called on every plan-cache miss, before the layers' packed weights are looked up.  A miss does not mean the
        weights changed visibly -- the first inverse call, a batch that crosses the 16-sample-tile threshold, an evicted
        plan -- and the per-layer packs are keyed on version counters, storage pointers and the `.data` salt: a write that
        announced itself through none of them since the last pack would be packed into the NEW plan from the OLD blobs.
        So: a layer whose key is the one its checksum was recorded under is COMPARED now (StalePackedWeights on a
        difference); a layer with a new key gets its checksum recorded."""
        from . import coupling
        if not coupling.VERIFY_WEIGHTS_EVERY:
            return
        flat = self.__dict__.get("_flat")
        if flat is None:
            return
        params = flat[3]
        if coupling._capturing(params):
            return
        versions, ptrs = tuple([p._version for p in params]), tuple([p.data_ptr() for p in params])
        for (c, _), (lo, hi) in zip(self, flat[4]):
            coupling._track_contents(c, "_contents_run", (flat[0], versions[lo:hi], ptrs[lo:hi]), params[lo:hi],
                                     periodic=False, compare_now=True)


def _permutation_key(p):
    perm = p._buffers["_permutation"]   # (the registered buffer, without nn.Module's attribute fallback)
    return id(perm), perm._version


def _run_weights_fingerprint(units):
    """What `_run_plan` keys the run's packed weights on: `_Run.weights_fingerprint()` -- one flat pass over the run --
    or, for a plain list of units, the per-layer keys."""
    if isinstance(units, _Run):
        return units.weights_fingerprint()
    from .coupling import _weights_key
    return tuple([_weights_key(c, c.transform_net) for c, _ in units])


def _run_geometry(units):
    return units.geometry() if isinstance(units, _Run) else units[0][0]._fused_geometry(tuple(c for c, _ in units[1:]))


class CompositeTransform(Transform):
    """Applies transforms in the given order; log-determinants add up (base.py:45-52).

    Two levels of fusion on 2-D inputs (`fuse_permutations`): a column Permutation next to a
    coupling layer is folded into that layer's kernel, and a run of consecutive
    [Permutation, spline coupling] pairs whose conditioners the whole-layer kernel (K8) covers is
    executed in ONE launch (`fuse_layer_runs`): rows are independent across the whole run, so the
    kernel walks its rows through every layer without leaving LDS."""

    fuse_layer_runs = True  # class-level switch for A/B measurements

    def __init__(self, transforms, fuse_permutations=True):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)
        self.fuse_permutations = fuse_permutations

    # ------------------------------------------------------------------ runs of K8 layers
    @staticmethod
    def _run_signature(coupling):
        return coupling._run_signature()

    def _collect_run(self, layers, start, inputs, context, inverse):
        """`_plan_run` behind a cache: which layers form a run depends on the call's shape (feature count, dtype,
        context, direction, grad mode), on the A/B switches and on per-layer attributes (`_run_signature`,
        training mode); the first two make the key, the last are re-read on every call (cheap attribute reads)
        and compared with what the cached plan saw.  Planning itself costs ~3 us per layer -- a 64-layer
        composite would spend more host time on it than the kernel takes on a small batch."""
        if not (self.fuse_layer_runs and inputs.dim() == 2 and inputs.shape[0] >= 1
                and inputs.dtype == torch.float32):
            return [], start
        ctx_key = None if context is None else (context.dim(), tuple(context.shape[1:]), context.dtype, context.is_cuda)
        key = (start, len(layers), inverse, inputs.shape[1], ctx_key, torch.is_grad_enabled(), _switch_state(),
               _cache.epoch())
        cache = self.__dict__.setdefault("_run_cache", {})
        hit = cache.get(key)
        if hit is not None:
            units, after, watched, seen = hit
            if len(layers) == len(watched) and all(a is b for a, b in zip(layers, watched)) \
                    and seen == self._watched_state(units, layers, after, inputs.shape[1], context):
                return units, after
        units, after = self._plan_run(layers, start, inputs, context, inverse)
        if units:   # (only runs are kept: "no run here" is decided afresh on every call)
            units = _Run(units)
            if len(cache) > 16:
                cache.clear()
            cache[key] = (units, after, list(layers), self._watched_state(units, layers, after, inputs.shape[1], context))
        return units, after

    @staticmethod
    def _joinable(t, features, context):
        kind = getattr(t, "_run_kind", None)   # whole-layer kernel this layer can join a run of
        # (a user's subclass, mixin or late assignment that overrides one of the reference's hooks joins none: coupling.py: _user_hooks)
        return (kind is not None and not t._user_hooks and t.unconditional_transform is None and t.features == features
                and kind(context) is not None)

    def _watched_state(self, units, layers, after, features, context):
        """What a cached plan is compared with on every call: the state of its layers and of the (up to two)
        layers behind it -- the run may have ended at one of them for a reason that no longer holds."""
        behind = [(t._run_signature(), self._joinable(t, features, context)) if getattr(t, "_run_kind", None) is not None else None
                  for t in layers[after:after + 2]]
        return _layer_state(units), behind

    def _plan_run(self, layers, start, inputs, context, inverse):
        """Longest run of units starting at `start`: forward a unit is [column Permutation]? +
        eligible coupling, inverse (layers already reversed) eligible coupling + [Permutation]?.
        Returns (units, next_index) with units = [(coupling, permutation or None)]."""
        units = []
        if not (self.fuse_layer_runs and inputs.dim() == 2 and inputs.shape[0] >= 1
                and inputs.dtype == torch.float32):
            return units, start

        def eligible(t):
            return self._joinable(t, inputs.shape[1], context)
        i, signature = start, None
        while i < len(layers):
            perm = None
            if not inverse:
                if _is_column_permutation(layers[i]) and i + 1 < len(layers) and eligible(layers[i + 1]):
                    perm, coupling, step = layers[i], layers[i + 1], 2
                elif eligible(layers[i]):
                    coupling, step = layers[i], 1
                else:
                    break
            else:
                if not eligible(layers[i]):
                    break
                coupling, step = layers[i], 1
                if i + 1 < len(layers) and _is_column_permutation(layers[i + 1]):
                    perm, step = layers[i + 1], 2
            sig = self._run_signature(coupling)
            if signature is not None and sig != signature:
                break
            signature = sig
            units.append((coupling, perm))
            i += step
        if len(units) < 2:
            return [], start
        geometry = _run_geometry(units)
        ce = getattr(getattr(units[0][0], "transform_net", None), "context_features", None) or 0
        if geometry[0] > 128 or geometry[1] > 64 or geometry[2] + ce > 64:   # the run's padded geometry (ops.fused_geometry)
            return [], start
        return units, i

    def _run_plan(self, units, inverse, tile16=False):
        """Concatenated weight / bias blobs and the composed tables of a run, cached until a weight or a
        permutation changes.  (weights, biases, tables, f16 stream or None, K8x's (weights, biases, scales) or None)."""
        from .. import ops
        first = units[0][0]
        mlp = type(first).__name__ in ("AffineCouplingTransform", "AdditiveCouplingTransform")
        geometry = _run_geometry(units)   # one padded geometry for the run
        f16 = (not mlp) and first._use_f16(geometry)
        x3 = (not mlp) and first._use_f16x3(geometry)     # K8x: three f16 pieces per operand (engine "f16x3")
        # (the key reads version counters only; the layers' packed blobs are looked at on a miss)
        tile16 = tile16 if f16 else 0
        ids = units.__dict__.get("_ids") if isinstance(units, _Run) else None
        if ids is None:
            ids = tuple([id(c) for c, _ in units])
            if isinstance(units, _Run):
                units.__dict__["_ids"] = ids
        key = (inverse, f16, x3, tile16, geometry, first._log2e() if not mlp else None, first.conditioner_act_scale if f16 else None,
               ids, _run_weights_fingerprint(units),
               tuple([None if p is None else _permutation_key(p) for _, p in units]))
        cache = self.__dict__.setdefault("_run_plans", {})
        plan = cache.get(key)
        if plan is None:
            if len(cache) > 4:
                cache.clear()
            if isinstance(units, _Run):
                units.verify_before_packing()
            packed = [c._packed_mlp() if mlp else c._packed_resnet(geometry) for c, _ in units]
            packed_f16 = [c._packed_resnet_f16(geometry, tile16) for c, _ in units] if f16 else None
            weights = torch.cat([w for w, _ in packed], dim=0).contiguous()
            biases = torch.cat([b for _, b in packed]).contiguous()
            spec_layers = []
            for c, p in units:
                perm = None if p is None else p._permutation
                spec_layers.append((c.transform_features, c.identity_features,
                                    None if inverse else perm, perm if inverse else None))
            Dp, dt4, di_u, _ = geometry
            tables = ops.flow_layer_tables(first.features, spec_layers, padded_features=Dp,
                                           padded_transform=dt4 if not mlp else None,
                                           padded_identity=di_u if not mlp else None)
            plan_f16 = ops.build_f16_stream(packed_f16, tables) if f16 else None
            plan_x3 = None
            if x3:
                packed_x3 = [c._packed_resnet_f16x3(geometry) for c, _ in units]
                plan_x3 = tuple(torch.cat([p[i] for p in packed_x3], dim=0).contiguous() for i in range(3))
            plan = (weights, biases, tables, plan_f16, plan_x3)
            cache[key] = plan
        return plan

    def _run_fused(self, units, inputs, total, context, inverse, standard_normal_log_prob=False):
        """One launch for the run (ragged batches are padded to full 128-row blocks inside `ops`).  Returns
        the outputs (or log_prob with `standard_normal_log_prob`), or None when the kernel declines."""
        from .. import ops
        first = units[0][0]
        for _, p in units:
            if p is not None:
                p._check(inputs)
        # (batches that give a CU at most one 128-row block: the 16-sample-tile kernel K8s and its own stream)
        act = first._block_activation() if hasattr(first, "_block_activation") else 0
        tile16 = (hasattr(first, "_use_f16") and inputs.is_cuda
                  and ops.use_tile16(inputs.shape[0], getattr(first, "num_bins", 0), context, inputs.device, act))
        weights, biases, tables, plan_f16, plan_x3 = self._run_plan(units, inverse, tile16)
        Dp, dt4, di_u, pad_value = _run_geometry(units)
        pad = (Dp, pad_value)
        if plan_x3 is not None:
            head = ops.rqs_coupling_resnet_f16x3(
                inputs, plan_x3, (weights, biases), tables, dt4, di_u, len(first.transform_net.blocks), first._spec(),
                inverse, total, num_layers=len(units), standard_normal_log_prob=standard_normal_log_prob, pad=pad)
            if head is not None:
                return head[1] if standard_normal_log_prob else head[0]
        if type(first).__name__ in ("AffineCouplingTransform", "AdditiveCouplingTransform"):
            hidden_linears, residual_blocks = first._conditioner_shape()
            head = ops.affine_flow_mlp(
                inputs, weights, biases, tables, first.num_transform_features, first.num_identity_features,
                hidden_linears, first._activation_code(), inverse, total,
                num_layers=len(units), standard_normal_log_prob=standard_normal_log_prob, pad=pad,
                residual_blocks=residual_blocks)
        elif plan_f16 is not None:
            head = ops.rqs_coupling_resnet_f16(
                inputs, plan_f16, (weights, biases), tables, dt4,
                di_u, len(first.transform_net.blocks), first._spec(), inverse,
                total, num_layers=len(units), standard_normal_log_prob=standard_normal_log_prob, pad=pad,
                context=context, tile16=tile16 if first._use_f16(_run_geometry(units)) else 0, activation=act)
            if head is None and tile16:
                # K8s declined (its ring + 16-row buffers + two copies of the parameter words exceed the LDS budget:
                # about six blocks at D = 128): K8h takes these shapes -- its own stream, the same call
                weights, biases, tables, plan_f16, _ = self._run_plan(units, inverse, False)
                if plan_f16 is not None:
                    head = ops.rqs_coupling_resnet_f16(
                        inputs, plan_f16, (weights, biases), tables, dt4, di_u, len(first.transform_net.blocks),
                        first._spec(), inverse, total, num_layers=len(units),
                        standard_normal_log_prob=standard_normal_log_prob, pad=pad, context=context, tile16=False,
                        activation=act)
        else:
            head = ops.rqs_coupling_resnet(
                inputs, weights, biases, tables, dt4, di_u,
                len(first.transform_net.blocks), first._spec(), inverse, total,
                log2e=first._log2e() if hasattr(first, "_log2e") else False, num_layers=len(units),
                standard_normal_log_prob=standard_normal_log_prob, context=context, pad=pad, activation=act)
        if head is None:
            return None
        return head[1] if standard_normal_log_prob else head[0]

    def standard_normal_log_prob(self, inputs, context=None):
        """Flow.log_prob (flows/base.py:42-49) for a StandardNormal base when the whole composite is ONE
        run of whole-layer kernels: the last layer's
        kernel adds -0.5 sum z^2 - 0.5 D log(2 pi) to the log-determinant while the rows are still
        on the chip, and z is never written.  Returns log_prob [batch], or None when the composite
        does not have that shape (the caller then takes the general route)."""
        if not self.fuse_permutations or inputs.dim() != 2:
            return None
        layers = list(self._transforms)
        units, after = self._collect_run(layers, 0, inputs, context, inverse=False)
        if not units or after != len(layers):
            return None
        return self._run_fused(units, inputs, None, context, inverse=False, standard_normal_log_prob=True)

    @staticmethod
    def _cascade(inputs, funcs, context):
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        for func in funcs:
            outputs, logabsdet = func(outputs, context)
            total += logabsdet
        return outputs, total

    def forward(self, inputs, context=None):
        layers = list(self._transforms)
        if not self.fuse_permutations:
            return self._cascade(inputs, layers, context)
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        i = 0
        while i < len(layers):
            units, after = self._collect_run(layers, i, outputs, context, inverse=False)
            if units:
                fused = self._run_fused(units, outputs, total, context, inverse=False)
                if fused is not None:
                    outputs, i = fused, after
                    continue
            t = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if nxt is not None and _is_column_permutation(t) and _accepts_fused_permutation(nxt, outputs):
                t._check(outputs)
                # permutation folded into the layer's gather, `total += logabsdet` into its kernel
                outputs, _ = nxt.forward(outputs, context, in_perm=t._permutation,
                                         logabsdet_accumulator=total)
                i += 2
            elif _accepts_fused_permutation(t, outputs):
                outputs, _ = t.forward(outputs, context, logabsdet_accumulator=total)
                i += 1
            else:
                outputs, logabsdet = t(outputs, context)
                total += logabsdet
                i += 1
        return outputs, total

    def inverse(self, inputs, context=None):
        layers = list(self._transforms)[::-1]
        if not self.fuse_permutations:
            return self._cascade(inputs, (t.inverse for t in layers), context)
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        i = 0
        while i < len(layers):
            units, after = self._collect_run(layers, i, outputs, context, inverse=True)
            if units:
                fused = self._run_fused(units, outputs, total, context, inverse=True)
                if fused is not None:
                    outputs, i = fused, after
                    continue
            t = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if nxt is not None and _is_column_permutation(nxt) and _accepts_fused_permutation(t, outputs):
                nxt._check(outputs)
                # Permutation.inverse after the layer == scatter through the forward permutation
                outputs, _ = t.inverse(outputs, context, out_scatter=nxt._permutation,
                                       logabsdet_accumulator=total)
                i += 2
            elif _accepts_fused_permutation(t, outputs):
                outputs, _ = t.inverse(outputs, context, logabsdet_accumulator=total)
                i += 1
            else:
                outputs, logabsdet = t.inverse(outputs, context)
                total += logabsdet
                i += 1
        return outputs, total


class MultiscaleCompositeTransform(Transform):
    """Real NVP's multiscale architecture (nflows/transforms/base.py:63-212): after every transform
    but the last the result is halved along `split_dim`; the first half leaves as output (flattened),
    the second half feeds the next transform.  Bookkeeping only -- the transforms added do the work.
    Outputs are always [batch, total]."""

    def __init__(self, num_transforms, split_dim=1):
        if not (isinstance(split_dim, int) and split_dim > 0):
            raise TypeError("Split dimension must be a positive integer.")
        super().__init__()
        self._transforms = nn.ModuleList()
        self._output_shapes = []
        self._num_transforms = num_transforms
        self._split_dim = split_dim

    def add_transform(self, transform, transform_output_shape):
        """To be called `num_transforms` times.  `transform_output_shape`: the shape of one sample
        leaving `transform`.  Returns the shape that continues to the next transform (None after the
        last one)."""
        if len(self._transforms) == self._num_transforms:
            raise RuntimeError("Adding more than {} transforms is not allowed.".format(self._num_transforms))
        axis = self._split_dim - 1
        if axis >= len(transform_output_shape):
            raise ValueError("No split_dim in output shape")
        if transform_output_shape[axis] < 2:
            raise ValueError("Size of dimension {} must be at least 2.".format(self._split_dim))
        self._transforms.append(transform)
        if len(self._transforms) == self._num_transforms:   # the last transform's result is not split
            self._output_shapes.append(tuple(transform_output_shape))
            return None
        size = transform_output_shape[axis]
        leaves, continues = list(transform_output_shape), list(transform_output_shape)
        leaves[axis] = (size + 1) // 2     # torch.chunk gives the larger half first
        continues[axis] = size // 2
        self._output_shapes.append(tuple(leaves))
        return tuple(continues)

    def _require_complete(self):
        if self._num_transforms != len(self._transforms):
            raise RuntimeError("Expecting exactly {} transform(s) to be added.".format(self._num_transforms))

    def forward(self, inputs, context=None):
        if self._split_dim >= inputs.dim():
            raise ValueError("No split_dim in inputs.")
        self._require_complete()
        batch = inputs.shape[0]
        total = inputs.new_zeros(batch)
        pieces = []
        hiddens = inputs
        last = len(self._transforms) - 1
        for i, transform in enumerate(self._transforms):
            result, logabsdet = transform(hiddens, context)
            total = total + logabsdet
            if i < last:
                out, hiddens = torch.chunk(result, chunks=2, dim=self._split_dim)
                assert tuple(out.shape[1:]) == self._output_shapes[i]
            else:
                out = result
            pieces.append(out.reshape(batch, -1))
        return torch.cat(pieces, dim=-1), total

    def inverse(self, inputs, context=None):
        if inputs.dim() != 2:
            raise ValueError("Expecting NxD inputs")
        self._require_complete()
        batch = inputs.shape[0]
        chunks, start = [], 0
        for shape in self._output_shapes:
            n = 1
            for d in shape:
                n *= d
            chunks.append(inputs[:, start:start + n].reshape(batch, *shape))
            start += n
        total = inputs.new_zeros(batch)
        hiddens, logabsdet = self._transforms[-1].inverse(chunks[-1], context)
        total = total + logabsdet
        for transform, chunk in zip(reversed(self._transforms[:-1]), reversed(chunks[:-1])):
            hiddens, logabsdet = transform.inverse(torch.cat((chunk, hiddens), dim=self._split_dim), context)
            total = total + logabsdet
        return hiddens, total


class InverseTransform(Transform):
    """Swaps forward and inverse of a transform (base.py:215-231)."""

    def __init__(self, transform):
        super().__init__()
        self._transform = transform

    def forward(self, inputs, context=None):
        return self._transform.inverse(inputs, context)

    def inverse(self, inputs, context=None):
        return self._transform(inputs, context)
