"""Exception types of the Transform API (reference: nflows/transforms/base.py:10-19)."""


class InverseNotAvailable(Exception):
    """Raised by a transform that cannot be inverted."""


class InputOutsideDomain(Exception):
    """Raised when an input lies outside the domain of a transform."""
