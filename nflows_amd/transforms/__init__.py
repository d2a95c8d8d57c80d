from .base import (CompositeTransform, InputOutsideDomain, InverseNotAvailable, InverseTransform,
                   Transform)
from .coupling import (AdditiveCouplingTransform, AffineCouplingTransform, CouplingTransform,
                       PiecewiseCubicCouplingTransform, PiecewiseLinearCouplingTransform,
                       PiecewiseQuadraticCouplingTransform,
                       PiecewiseRationalQuadraticCouplingTransform)
from .permutations import Permutation, RandomPermutation, ReversePermutation
from . import splines
from .autoregressive import (AutoregressiveTransform, MaskedAffineAutoregressiveTransform,
                             MaskedPiecewiseCubicAutoregressiveTransform,
                             MaskedPiecewiseLinearAutoregressiveTransform,
                             MaskedPiecewiseQuadraticAutoregressiveTransform,
                             MaskedPiecewiseRationalQuadraticAutoregressiveTransform)
from .made import MADE
from .nonlinearities import (PiecewiseCubicCDF, PiecewiseLinearCDF, PiecewiseQuadraticCDF,
                            PiecewiseRationalQuadraticCDF)
