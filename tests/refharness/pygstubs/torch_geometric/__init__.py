"""functional stand-in for the few torch_geometric names the reference's Decima code uses
(test infrastructure)"""
from . import data, utils  # noqa: F401
