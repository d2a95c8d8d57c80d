from .base import (CompositeTransform, InputOutsideDomain, InverseNotAvailable, InverseTransform,
                   MultiscaleCompositeTransform, Transform)
from .standard import AffineScalarTransform, AffineTransform, IdentityTransform, PointwiseAffineTransform
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
