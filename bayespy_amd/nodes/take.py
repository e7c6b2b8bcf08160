"""
Deterministic nodes that re-arrange plates: Take, Concatenate, Gate (reference:
bayespy/inference/vmp/nodes/take.py:15-140, concatenate.py:15-167, gate.py:20-205).

They pass the moments of their parent(s) on unchanged in kind (``dims`` are the parent's),
only the plates differ:

* ``Take(node, indices, plate_axis=-1)`` -- ``np.take`` along a plate axis (an index array with
  several axes creates several plate axes);
* ``Concatenate(*nodes, axis=-1)`` -- ``np.concatenate`` along a plate axis;
* ``Gate(Z, X, gated_plate=-1)`` -- the plate axis ``gated_plate`` of ``X`` is averaged with the
  class probabilities of the categorical ``Z``;
* ``Slice(X, index)`` = ``X[index]`` -- basic indexing of the plates with integers, slices,
  ``None`` and ``...`` (node.py:761-763, :868-1130);
* ``Choose(z, *nodes)`` -- ``Gate`` over the concatenation of the nodes (gate.py:207-250).
"""
import numpy as np

from .node import Node, Constant
from ..utils.shapes import broadcasted_shape


def _kind_of(parent):
    from .gaussian import _is_gaussian
    return _is_gaussian(parent)


class Take(Node):

    def __init__(self, node, indices, plate_axis=-1, name=None):
        if not isinstance(node, Node) or isinstance(node, Constant):
            raise ValueError("Take needs a node as its first argument")
        if not isinstance(plate_axis, (int, np.integer)):
            raise ValueError("Plate axis must be integer")
        if plate_axis >= 0:
            raise ValueError("plate_axis must be negative index")
        if plate_axis < -len(node.plates):
            raise ValueError("plate_axis out of bounds")
        idx = np.array(indices)
        if not np.issubdtype(idx.dtype, np.integer):
            raise ValueError("Indices must be integers")
        length = node.plates[plate_axis]
        if np.any(idx < -length) or np.any(idx >= length):
            raise ValueError("Index out of bounds")
        plates = node.plates[:plate_axis] + idx.shape
        if plate_axis != -1:
            plates = plates + node.plates[plate_axis + 1:]
        super().__init__(node, plates=plates, dims=node.dims, name=name)
        self.indices = idx
        self.plate_axis = int(plate_axis)
        self.original_length = int(length)
        self._gaussian_like = _kind_of(node)
        for attr in ('shape', 'ndim'):
            if hasattr(node, attr):
                setattr(self, attr, getattr(node, attr))


class Concatenate(Node):

    def __init__(self, *nodes, axis=-1, name=None):
        if axis >= 0:
            raise ValueError("Currently, only negative axis indeces are allowed.")
        if len(nodes) < 1 or not all(isinstance(n, Node) and not isinstance(n, Constant)
                                     for n in nodes):
            raise ValueError("Couldn't determine parent moments")
        dims = tuple(nodes[0].dims)
        for n in nodes:
            if tuple(n.dims) != dims:
                raise ValueError("Parents have different dimensionalities")
            if len(n.plates) < -axis:
                raise ValueError("A parent does not have the plate axis %d" % axis)
        others = []
        for n in nodes:
            p = list(n.plates)
            p[axis] = 1
            others.append(tuple(p))
        try:
            common = list(broadcasted_shape(*others))
        except ValueError:
            raise ValueError("Plates of the parents differ on other axes than the concatenated one")
        self.lengths = [int(n.plates[axis]) for n in nodes]
        common[axis] = int(sum(self.lengths))
        super().__init__(*nodes, plates=tuple(common), dims=dims, name=name)
        self.axis = int(axis)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        self._gaussian_like = _kind_of(nodes[0])
        for attr in ('shape', 'ndim'):
            if hasattr(nodes[0], attr):
                setattr(self, attr, getattr(nodes[0], attr))


class Gate(Node):

    def __init__(self, Z, X, gated_plate=-1, name=None):
        if gated_plate >= 0:
            raise ValueError("Cluster plate must be negative integer")
        if not isinstance(X, Node) or isinstance(X, Constant):
            raise ValueError("X must be a node or moments should be provided")
        if len(X.plates) < abs(gated_plate):
            raise ValueError("The gated node does not have a plate axis is gated")
        K = X.plates[gated_plate]
        if hasattr(Z, 'as_categorical'):
            Z = Z.as_categorical()
        super().__init__(Z, X, plates=(), dims=X.dims, name=name)
        z = self.parents[0]
        if isinstance(z, Constant):
            # fixed class labels (CategoricalMoments.compute_fixed_moments, categorical.py:30-46)
            if np.any(z.value != np.round(z.value)):
                raise ValueError("Values must be integers")
            if np.any(z.value < 0) or np.any(z.value >= K):
                raise ValueError("Invalid category index")
            zplates = z.value.shape
        else:
            if tuple(z.dims) != ((K,),):
                raise ValueError("Inconsistent number of clusters")
            zplates = z.plates
        xp = list(X.plates)
        xp.pop(gated_plate)
        self.plates = broadcasted_shape(zplates, tuple(xp))
        self.gated_plate = int(gated_plate)
        self.K = int(K)
        self._gaussian_like = _kind_of(X)
        for attr in ('shape', 'ndim'):
            if hasattr(X, attr):
                setattr(self, attr, getattr(X, attr))


def _slicelen(s):
    return len(range(s.start, s.stop, s.step))


class Slice(Node):
    """Basic slicing of the plates: ``X[2:5]``, ``X[..., 0]``, ``X[:, None]`` ..."""

    def __init__(self, X, slices, name=None):
        if not isinstance(X, Node) or isinstance(X, Constant):
            raise ValueError("Slice needs a node")
        slices = list(slices) if isinstance(slices, tuple) else [slices]
        num_axis, ellipsis_index = 0, None
        for k, sl in enumerate(slices):
            if isinstance(sl, (int, np.integer)) or isinstance(sl, slice):
                num_axis += 1
            elif sl is None:
                pass
            elif sl is Ellipsis:
                if ellipsis_index is None:
                    ellipsis_index = k
                else:                                  # later ellipses stand for ":"
                    num_axis += 1
                    slices[k] = slice(None)
            else:
                raise TypeError("Invalid argument type: {0}".format(sl.__class__))
        if num_axis > len(X.plates):
            raise IndexError("Too many indices")
        expand = [slice(None)] * (len(X.plates) - num_axis)
        if ellipsis_index is not None:
            slices = slices[:ellipsis_index] + expand + slices[ellipsis_index + 1:]
        else:
            slices = slices + expand
        j = 0
        plates = []
        for k, sl in enumerate(slices):
            if isinstance(sl, (int, np.integer)):
                sl = int(sl)
                if sl < 0:
                    sl += X.plates[j]
                if sl < 0 or sl >= X.plates[j]:
                    raise IndexError("Index out of range")
                slices[k] = sl
                j += 1
            elif isinstance(sl, slice):
                sl = slice(*sl.indices(X.plates[j]))
                if _slicelen(sl) <= 0:
                    raise IndexError("Slicing leads to empty plates")
                slices[k] = sl
                plates.append(_slicelen(sl))
                j += 1
            else:
                plates.append(1)
        super().__init__(X, plates=tuple(plates), dims=X.dims, name=name)
        self.slices = slices
        self._gaussian_like = _kind_of(X)
        for attr in ('shape', 'ndim'):
            if hasattr(X, attr):
                setattr(self, attr, getattr(X, attr))


def Choose(z, *nodes):
    """Choose plate elements from ``nodes`` by the categorical variable ``z``: a gate over the
    concatenation of the nodes along a new last plate axis (gate.py:207-250)."""
    combined = Concatenate(*[n[..., None] for n in nodes])
    return Gate(z, combined)
