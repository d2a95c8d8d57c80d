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
_epoch = 0


def epoch():
    return _epoch


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
