"""``bayespy_amd.inference`` -- mirrors ``bayespy.inference`` (bayespy/inference/__init__.py:34)."""
from .vb import VB

__all__ = ['VB']
