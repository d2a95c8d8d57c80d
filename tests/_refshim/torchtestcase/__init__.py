"""Test-side stub for the un-vendored PyPI package `torchtestcase` (reference tests only)."""
import unittest

import torch


class TorchTestCase(unittest.TestCase):
    _eps = 0.0

    @property
    def eps(self):
        return self._eps

    @eps.setter
    def eps(self, value):
        self._eps = float(value)

    def _fail_with_message(self, msg, standard_msg):
        self.fail(self._formatMessage(msg, standard_msg))

    def assertEqual(self, first, second, msg=None):
        if torch.is_tensor(first) and torch.is_tensor(second):
            if first.shape != second.shape:
                self._fail_with_message(msg, "shapes differ: %s vs %s" % (first.shape, second.shape))
            if self._eps:
                bad = (first.double() - second.double()).abs().max().item() >= self._eps if first.numel() else False
            else:
                bad = not torch.equal(first, second)
            if bad:
                self._fail_with_message(msg, "tensors differ")
        else:
            super().assertEqual(first, second, msg)

    def assert_tensor_less_equal(self, first, second, msg=None):
        if not bool((torch.as_tensor(first) <= torch.as_tensor(second)).all()):
            self._fail_with_message(msg, "not <=")

    def assert_tensor_greater_equal(self, first, second, msg=None):
        if not bool((torch.as_tensor(first) >= torch.as_tensor(second)).all()):
            self._fail_with_message(msg, "not >=")
