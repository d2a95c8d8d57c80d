"""Flow = invertible transform + base distribution (API of nflows/flows/base.py:12-120).

The caller of the hot path: `log_prob` = transform forward + base log-density, `sample` = base
draw + transform inverse.  With a `StandardNormal` base on the GPU the base density and the final
`+ logabsdet` (flows/base.py:49) are one kernel.
"""
from inspect import signature

import torch
from torch import nn

from .. import ops
from ..distributions.base import Distribution
from ..distributions.normal import StandardNormal
from ..utils import torchutils


class Flow(Distribution):
    def __init__(self, transform, distribution, embedding_net=None):
        super().__init__()
        self._transform = transform
        self._distribution = distribution
        self._context_used_in_base = "context" in signature(distribution.log_prob).parameters
        if embedding_net is None:
            embedding_net = nn.Identity()
        else:
            assert isinstance(embedding_net, nn.Module), (
                "embedding_net is not a nn.Module. If you want to use hard-coded summary features, "
                "please simply pass the encoded features and pass embedding_net=None")
        self._embedding_net = embedding_net

    # -- density ---------------------------------------------------------------------------
    def _base_log_prob(self, noise, embedded):
        if self._context_used_in_base:
            return self._distribution.log_prob(noise, context=embedded)
        return self._distribution.log_prob(noise)

    def _log_prob(self, inputs, context):
        embedded = self._embedding_net(context)
        base = self._distribution
        fused = getattr(self._transform, "standard_normal_log_prob", None)
        if (fused is not None and type(base) is StandardNormal and inputs.is_cuda
                and inputs.shape[1:] == base._shape):
            log_prob = fused(inputs, embedded)   # density folded into the last layer's kernel
            if log_prob is not None:
                return log_prob
        noise, logabsdet = self._transform(inputs, context=embedded)
        if type(base) is StandardNormal and noise.is_cuda:
            if noise.shape[1:] != base._shape:
                raise ValueError("Expected input of shape {}, got {}".format(base._shape, noise.shape[1:]))
            return ops.standard_normal_log_prob(noise, logabsdet)
        return self._base_log_prob(noise, embedded) + logabsdet

    # -- sampling --------------------------------------------------------------------------
    def _noise(self, num_samples, embedded, with_log_prob):
        draw = self._distribution.sample_and_log_prob if with_log_prob else self._distribution.sample
        if self._context_used_in_base:
            return draw(num_samples, context=embedded)
        if with_log_prob or embedded is None:
            return draw(num_samples)
        flat = draw(num_samples * embedded.shape[0])
        return flat.reshape(embedded.shape[0], -1, flat.shape[1])

    def _through_inverse(self, noise, embedded, num_samples):
        """Runs the inverse on [context * samples, ...] rows and restores the context dimension."""
        if embedded is None:
            return self._transform.inverse(noise, context=None)
        rows = torchutils.merge_leading_dims(noise, num_dims=2)
        ctx = torchutils.repeat_rows(embedded, num_reps=num_samples)
        samples, logabsdet = self._transform.inverse(rows, context=ctx)
        grouped = [-1, num_samples]
        return (torchutils.split_leading_dim(samples, shape=grouped),
                torchutils.split_leading_dim(logabsdet, shape=grouped))

    def _sample(self, num_samples, context):
        embedded = self._embedding_net(context)
        noise = self._noise(num_samples, embedded, with_log_prob=False)
        return self._through_inverse(noise, embedded, num_samples)[0]

    def sample_and_log_prob(self, num_samples, context=None):
        """Samples and their log-densities from one inverse pass (flows/base.py:77-106)."""
        embedded = self._embedding_net(context)
        noise, base_log_prob = self._noise(num_samples, embedded, with_log_prob=True)
        samples, logabsdet = self._through_inverse(noise, embedded, num_samples)
        return samples, base_log_prob - logabsdet

    def transform_to_noise(self, inputs, context=None):
        """Data -> noise (flows/base.py:108-120)."""
        return self._transform(inputs, context=self._embedding_net(context))[0]
