from . import nets
