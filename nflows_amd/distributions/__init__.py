from .base import Distribution, NoMeanException
from .normal import StandardNormal
