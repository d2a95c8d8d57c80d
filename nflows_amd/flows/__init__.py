from .base import Flow
