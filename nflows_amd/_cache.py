"""Epoch of the packed-weight caches.

The whole-layer kernels read re-tiled copies of the conditioner weights (ops.pack_*), cached per
layer and keyed on the parameters' (data_ptr, _version).  Writes that bypass the version counter
(`p.data.copy_(...)`, `dist.broadcast(p.data)`, EMA weight swaps through `.data`) are invisible to
that key; every cache key therefore also carries this epoch, which `invalidate()` advances.
`nflows_amd.invalidate_packed_weights()` is the public name; `load_state_dict`, `.to()/.cuda()`
and `parallel.broadcast_model` call it themselves.
"""
_epoch = 0


def epoch():
    return _epoch


def invalidate():
    global _epoch
    _epoch += 1
