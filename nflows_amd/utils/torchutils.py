"""Host-side helpers of the hot path (reference: nflows/utils/torchutils.py).

Mask builders and leading-dimension reshapes are plain index bookkeeping and stay in Python;
`sum_except_batch` on a HIP float32 tensor is the K3 row-sum kernel.
"""
import torch

from . import typechecks as check


def sum_except_batch(x, num_batch_dims=1):
    """Sum over all but the first `num_batch_dims` dimensions (torchutils.py:19-24)."""
    if not check.is_nonnegative_int(num_batch_dims):
        raise TypeError("Number of batch dimensions must be a non-negative integer.")
    if num_batch_dims == 1 and x.dim() >= 1 and x.is_cuda and x.dtype == torch.float32 \
            and not (torch.is_grad_enabled() and x.requires_grad):
        from .. import ops
        return ops.rowsum(x)
    return torch.sum(x, dim=list(range(num_batch_dims, x.dim())))


def split_leading_dim(x, shape):
    """[prod(shape), ...] -> [*shape, ...] (torchutils.py:27-30)."""
    return x.reshape(torch.Size(shape) + x.shape[1:])


def merge_leading_dims(x, num_dims):
    """Collapses the first `num_dims` dimensions into one (torchutils.py:33-42)."""
    if not check.is_positive_int(num_dims):
        raise TypeError("Number of leading dims must be a positive integer.")
    if num_dims > x.dim():
        raise ValueError("Number of leading dims can't be greater than total number of dims.")
    return x.reshape(torch.Size([-1]) + x.shape[num_dims:])


def repeat_rows(x, num_reps):
    """Row i of x appears num_reps times consecutively (torchutils.py:45-52)."""
    if not check.is_positive_int(num_reps):
        raise TypeError("Number of repetitions must be a positive integer.")
    return torch.repeat_interleave(x, num_reps, dim=0)


def create_alternating_binary_mask(features, even=True):
    """uint8 mask 1,0,1,0,... (even=True) or 0,1,0,1,... (torchutils.py:89-100)."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[(0 if even else 1)::2] = 1
    return mask


def create_mid_split_binary_mask(features):
    """First ceil(features/2) entries are 1 (torchutils.py:103-113)."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[: (features + 1) // 2] = 1
    return mask


def create_random_binary_mask(features):
    """ceil(features/2) ones at positions drawn without replacement (torchutils.py:116-131)."""
    mask = torch.zeros(features, dtype=torch.uint8)
    picks = torch.multinomial(torch.ones(features), num_samples=(features + 1) // 2, replacement=False)
    mask[picks] += 1
    return mask


def searchsorted(bin_locations, inputs, eps=1e-6):
    """Bin index by counting knots <= input, last knot nudged by eps (torchutils.py:134-136).
    Float32 tensors on the device: the `nfa_searchsorted_f32` kernel (the spline kernels fuse the same
    search and can report its result: `return_bin_idx` of `ops.rqs_elementwise` / `ops.rqs_coupling`);
    anything else (host-side bookkeeping on small tensors, other dtypes): the same count in PyTorch.
    Unlike the reference it does not modify `bin_locations` in place."""
    if (torch.is_tensor(bin_locations) and torch.is_tensor(inputs) and inputs.is_cuda and bin_locations.is_cuda
            and inputs.dtype == torch.float32 and bin_locations.dtype == torch.float32
            and bin_locations.dim() >= 1):
        from .. import ops
        return ops.searchsorted(bin_locations, inputs, eps)
    knots = bin_locations.clone()
    knots[..., -1] += eps
    return (inputs[..., None] >= knots).sum(dim=-1) - 1
