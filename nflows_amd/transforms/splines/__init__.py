from .rational_quadratic import (DEFAULT_MIN_BIN_HEIGHT, DEFAULT_MIN_BIN_WIDTH,
                                 DEFAULT_MIN_DERIVATIVE, rational_quadratic_spline,
                                 unconstrained_rational_quadratic_spline)
from . import rational_quadratic
