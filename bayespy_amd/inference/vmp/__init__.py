"""``bayespy_amd.inference.vmp`` -- import-path mirror of ``bayespy.inference.vmp`` so that
``from bayespy.inference.vmp import transformations`` (demos/pca.py:16) ports by renaming
the top-level package only."""
from .. import transformations          # noqa: F401
from ... import nodes                   # noqa: F401
from ..vb import VB                     # noqa: F401
