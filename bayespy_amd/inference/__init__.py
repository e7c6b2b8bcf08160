"""``bayespy_amd.inference`` -- mirrors ``bayespy.inference`` (bayespy/inference/__init__.py:34)."""
from .vb import VB
from .plans.extension import register_family, unregister_family

__all__ = ['VB', 'register_family', 'unregister_family']
