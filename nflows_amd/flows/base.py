"""Flow = transform + base distribution (reference: nflows/flows/base.py:12-120).

The caller of the hot path: `log_prob` runs the transform forward and adds the base log-density;
`sample` draws noise and runs the transform inverse.  When the base distribution is a
`StandardNormal` the final `log_prob + logabsdet` add (flows/base.py:49) is fused into the
base-density kernel.
"""
from inspect import signature

import torch.nn

from .. import ops
from ..distributions.base import Distribution
from ..distributions.normal import StandardNormal
from ..utils import torchutils


class Flow(Distribution):
    def __init__(self, transform, distribution, embedding_net=None):
        super().__init__()
        self._transform = transform
        self._distribution = distribution
        self._context_used_in_base = "context" in signature(self._distribution.log_prob).parameters
        if embedding_net is not None:
            assert isinstance(embedding_net, torch.nn.Module), (
                "embedding_net is not a nn.Module. If you want to use hard-coded summary features, "
                "please simply pass the encoded features and pass embedding_net=None")
            self._embedding_net = embedding_net
        else:
            self._embedding_net = torch.nn.Identity()

    def _log_prob(self, inputs, context):
        embedded = self._embedding_net(context)
        noise, logabsdet = self._transform(inputs, context=embedded)
        if type(self._distribution) is StandardNormal and noise.is_cuda:
            if noise.shape[1:] != self._distribution._shape:
                raise ValueError("Expected input of shape {}, got {}".format(
                    self._distribution._shape, noise.shape[1:]))
            return ops.standard_normal_log_prob(noise, logabsdet)
        if self._context_used_in_base:
            log_prob = self._distribution.log_prob(noise, context=embedded)
        else:
            log_prob = self._distribution.log_prob(noise)
        return log_prob + logabsdet

    def _draw_noise(self, num_samples, embedded, with_log_prob):
        fn = self._distribution.sample_and_log_prob if with_log_prob else self._distribution.sample
        if self._context_used_in_base:
            return fn(num_samples, context=embedded)
        if with_log_prob or embedded is None:
            return fn(num_samples)
        flat = fn(num_samples * embedded.shape[0])
        return torch.reshape(flat, (embedded.shape[0], -1, flat.shape[1]))

    def _sample(self, num_samples, context):
        embedded = self._embedding_net(context)
        noise = self._draw_noise(num_samples, embedded, with_log_prob=False)
        if embedded is not None:
            noise = torchutils.merge_leading_dims(noise, num_dims=2)
            embedded = torchutils.repeat_rows(embedded, num_reps=num_samples)
        samples, _ = self._transform.inverse(noise, context=embedded)
        if embedded is not None:
            samples = torchutils.split_leading_dim(samples, shape=[-1, num_samples])
        return samples

    def sample_and_log_prob(self, num_samples, context=None):
        """Samples and their log-densities from one inverse pass (flows/base.py:77-106)."""
        embedded = self._embedding_net(context)
        noise, log_prob = self._draw_noise(num_samples, embedded, with_log_prob=True)
        if embedded is not None:
            noise = torchutils.merge_leading_dims(noise, num_dims=2)
            embedded = torchutils.repeat_rows(embedded, num_reps=num_samples)
        samples, logabsdet = self._transform.inverse(noise, context=embedded)
        if embedded is not None:
            samples = torchutils.split_leading_dim(samples, shape=[-1, num_samples])
            logabsdet = torchutils.split_leading_dim(logabsdet, shape=[-1, num_samples])
        return samples, log_prob - logabsdet

    def transform_to_noise(self, inputs, context=None):
        """Data -> noise (flows/base.py:108-120)."""
        noise, _ = self._transform(inputs, context=self._embedding_net(context))
        return noise
