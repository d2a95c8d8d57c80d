"""Small argument predicates (reference: nflows/utils/typechecks.py)."""
import numbers


def is_bool(x):
    return isinstance(x, bool)


def is_int(x):
    return isinstance(x, numbers.Integral) and not isinstance(x, bool)


def is_positive_int(x):
    return is_int(x) and x > 0


def is_nonnegative_int(x):
    return is_int(x) and x >= 0


def is_power_of_two(n):
    return is_positive_int(n) and (n & (n - 1)) == 0
