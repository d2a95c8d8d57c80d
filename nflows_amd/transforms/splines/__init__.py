from .rational_quadratic import (DEFAULT_MIN_BIN_HEIGHT, DEFAULT_MIN_BIN_WIDTH,
                                 DEFAULT_MIN_DERIVATIVE, rational_quadratic_spline,
                                 unconstrained_rational_quadratic_spline)
from . import rational_quadratic
from .linear import linear_spline, unconstrained_linear_spline
from .quadratic import quadratic_spline, unconstrained_quadratic_spline
from .cubic import cubic_spline, unconstrained_cubic_spline
from . import cubic, linear, quadratic
