"""Test-side stub for the un-vendored PyPI package `UMNN`.

The reference's `nflows/transforms/UMNN/MonotonicNormalizer.py:2` imports two names from it
at module import time; nothing on the coupling hot path uses them.  This stub only exists so
that `/root/reference` can be imported as a parity oracle when generating golden vectors.
It is NOT product code and is never imported by `nflows_amd`.
"""


class NeuralIntegral:  # pragma: no cover - placeholder
    pass


class ParallelNeuralIntegral:  # pragma: no cover - placeholder
    pass
