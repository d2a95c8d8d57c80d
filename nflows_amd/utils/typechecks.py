"""Argument predicates used by the host-side classes."""
import numbers


def is_bool(value):
    return type(value) is bool


def is_int(value):
    # bool is an Integral in Python; the checks below are about counts, so exclude it
    return isinstance(value, numbers.Integral) and not is_bool(value)


def _int_at_least(value, lowest):
    return is_int(value) and value >= lowest


def is_positive_int(value):
    return _int_at_least(value, 1)


def is_nonnegative_int(value):
    return _int_at_least(value, 0)


def is_power_of_two(value):
    return _int_at_least(value, 1) and value & (value - 1) == 0
