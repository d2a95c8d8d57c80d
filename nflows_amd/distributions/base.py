"""Distribution template methods used by Flow.

Public surface of nflows/distributions/base.py:16-128: `log_prob(inputs, context)`,
`sample(num_samples, context, batch_size)`, `sample_and_log_prob`, `mean`; subclasses implement
`_log_prob`, `_sample`, `_mean`.
"""
import torch
from torch import nn

from ..utils import torchutils
from ..utils import typechecks as check


class NoMeanException(Exception):
    """The distribution does not define a mean."""


def _as_tensor_or_none(value):
    return None if value is None else torch.as_tensor(value)


class Distribution(nn.Module):
    def forward(self, *args):
        raise RuntimeError("Forward method cannot be called for a Distribution object.")

    # -- density ---------------------------------------------------------------------------
    def log_prob(self, inputs, context=None):
        """[batch, ...] -> [batch]; one context row per input row when a context is given."""
        inputs, context = torch.as_tensor(inputs), _as_tensor_or_none(context)
        if context is not None and context.shape[0] != inputs.shape[0]:
            raise ValueError("Number of input items must be equal to number of context items.")
        return self._log_prob(inputs, context)

    def _log_prob(self, inputs, context):
        raise NotImplementedError()

    # -- sampling --------------------------------------------------------------------------
    def sample(self, num_samples, context=None, batch_size=None):
        """`num_samples` draws ([context_size, num_samples, ...] with a context); `batch_size`
        splits the work into chunks that are concatenated."""
        if not check.is_positive_int(num_samples):
            raise TypeError("Number of samples must be a positive integer.")
        context = _as_tensor_or_none(context)
        if batch_size is None:
            return self._sample(num_samples, context)
        if not check.is_positive_int(batch_size):
            raise TypeError("Batch size must be a positive integer.")
        sizes = [batch_size] * (num_samples // batch_size)
        if num_samples % batch_size:
            sizes.append(num_samples % batch_size)
        return torch.cat([self._sample(n, context) for n in sizes], dim=0)

    def _sample(self, num_samples, context):
        raise NotImplementedError()

    def sample_and_log_prob(self, num_samples, context=None):
        samples = self.sample(num_samples, context=context)
        if context is None:
            return samples, self.log_prob(samples)
        flat = torchutils.merge_leading_dims(samples, num_dims=2)
        rows = torchutils.repeat_rows(context, num_reps=num_samples)
        assert flat.shape[0] == rows.shape[0]
        log_prob = self.log_prob(flat, context=rows)
        return samples, torchutils.split_leading_dim(log_prob, shape=[-1, num_samples])

    # -- moments ---------------------------------------------------------------------------
    def mean(self, context=None):
        return self._mean(_as_tensor_or_none(context))

    def _mean(self, context):
        raise NoMeanException()
