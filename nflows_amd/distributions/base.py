"""Distribution template methods used by Flow (reference: nflows/distributions/base.py)."""
import torch
from torch import nn

from ..utils import torchutils
from ..utils import typechecks as check


class NoMeanException(Exception):
    """Raised when a distribution has no mean."""


class Distribution(nn.Module):
    """log_prob / sample / sample_and_log_prob wrappers around `_log_prob`, `_sample`."""

    def forward(self, *args):
        raise RuntimeError("Forward method cannot be called for a Distribution object.")

    def log_prob(self, inputs, context=None):
        """[batch, ...] -> [batch] log-densities; context rows must match inputs rows."""
        inputs = torch.as_tensor(inputs)
        if context is not None:
            context = torch.as_tensor(context)
            if inputs.shape[0] != context.shape[0]:
                raise ValueError("Number of input items must be equal to number of context items.")
        return self._log_prob(inputs, context)

    def _log_prob(self, inputs, context):
        raise NotImplementedError()

    def sample(self, num_samples, context=None, batch_size=None):
        """num_samples draws (per context row if a context is given), optionally in chunks."""
        if not check.is_positive_int(num_samples):
            raise TypeError("Number of samples must be a positive integer.")
        if context is not None:
            context = torch.as_tensor(context)
        if batch_size is None:
            return self._sample(num_samples, context)
        if not check.is_positive_int(batch_size):
            raise TypeError("Batch size must be a positive integer.")
        full, rest = divmod(num_samples, batch_size)
        chunks = [self._sample(batch_size, context) for _ in range(full)]
        if rest > 0:
            chunks.append(self._sample(rest, context))
        return torch.cat(chunks, dim=0)

    def _sample(self, num_samples, context):
        raise NotImplementedError()

    def sample_and_log_prob(self, num_samples, context=None):
        """Samples together with their log-densities."""
        samples = self.sample(num_samples, context=context)
        if context is not None:
            samples = torchutils.merge_leading_dims(samples, num_dims=2)
            context = torchutils.repeat_rows(context, num_reps=num_samples)
            assert samples.shape[0] == context.shape[0]
        log_prob = self.log_prob(samples, context=context)
        if context is not None:
            samples = torchutils.split_leading_dim(samples, shape=[-1, num_samples])
            log_prob = torchutils.split_leading_dim(log_prob, shape=[-1, num_samples])
        return samples, log_prob

    def mean(self, context=None):
        if context is not None:
            context = torch.as_tensor(context)
        return self._mean(context)

    def _mean(self, context):
        raise NoMeanException()
