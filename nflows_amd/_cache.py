"""Epoch of the packed-weight caches.

The whole-layer kernels read re-tiled copies of the conditioner weights (ops.pack_*), cached per
layer and keyed on the parameters' (data_ptr, _version).  Writes that bypass the version counter
(`p.data.copy_(...)`, `dist.broadcast(p.data)`, EMA weight swaps through `.data`) are invisible to
that key; every cache key therefore also carries this epoch, which `invalidate()` advances.
`nflows_amd.invalidate_packed_weights()` is the public name; `load_state_dict`, `.to()/.cuda()`
and `parallel.broadcast_model` call it themselves, and so does every registration of a Parameter or
buffer object on a module (global hooks below): the cache keys read the version counters of a LIST of a
conditioner's weights made once per epoch instead of walking the module tree on every call.
"""
import torch

_epoch = 0
_data_epoch = 0


def epoch():
    return _epoch


def data_epoch():
    """Advances whenever the `.data` of a WATCHED parameter is read or assigned (below): somebody may have written through
    it.  The weight keys compare the CONTENTS of their parameters with the checksum on record when this has moved since they
    last looked (transforms/coupling.py: _weights_key, transforms/base.py: _Run.weights_fingerprint) -- one batched
    comparison per event, and only layers whose contents really changed are repacked."""
    return _data_epoch


_base_data = torch._C.TensorBase.data


class WatchedParameter(torch.nn.Parameter):
    """A Parameter whose `.data` tells the packed-weight caches that it was touched (round 6).  A write through `.data`
    -- `p.data.copy_(ema)`, `dist.broadcast(p.data)`, `p.data.mul_(2)` -- reaches the storage without advancing the version
    counter the cache keys read: until round 5 the fused kernels then ran on the OLD packed weights until a periodic
    checksum noticed (up to 255 evaluations later, as an exception).  The reference reads `self.transform_net`'s
    parameters on every call (coupling.py:85); with this class the next call after such a write repacks from the new
    values: no stale evaluation, no exception, no device synchronisation on calls without such an event.  Parameters of
    the conditioners the fused kernels serve are re-classed in place (`watch`: same object, same identity for optimizers
    and DDP; pickling and state_dict are plain Parameter's).  Writes that avoid even this -- a view's `.data`, raw
    storage, foreign kernels on `data_ptr()` -- are left to the periodic checksum (NFA_VERIFY_WEIGHTS)."""

    @property
    def data(self):
        global _data_epoch
        _data_epoch += 1
        return _base_data.__get__(self, type(self))

    @data.setter
    def data(self, value):
        global _data_epoch
        _data_epoch += 1
        _base_data.__set__(self, value)


def watch(p):
    """re-class a plain Parameter as WatchedParameter, in place"""
    if type(p) is torch.nn.Parameter:
        p.__class__ = WatchedParameter


def invalidate():
    global _epoch
    _epoch += 1


def _on_registration(module, name, value):
    # a Parameter / buffer object (re)registered on any module: the per-layer lists of weights the cache keys
    # are built from (transforms/coupling.py: _weights_key) may be stale
    invalidate()


def _install_hooks():
    try:
        from torch.nn.modules import module as _m
        _m.register_module_parameter_registration_hook(_on_registration)
        _m.register_module_buffer_registration_hook(_on_registration)
    except (ImportError, AttributeError):  # (older torch: the keys fall back to walking the module tree)
        return False
    return True


HOOKED = _install_hooks()
