from . import functional, nets
