"""
Add node (reference: bayespy/inference/vmp/nodes/add.py:15-154).

``Add(X1, X2, ...)``: the sum of Gaussian-moment parents with identical variable shapes; the
plates broadcast.  The parents are independent under q, so
<ss^T> = sum_i <x_i x_i^T> + sum_{i != j} <x_i><x_j>^T  (add.py:95-108).
"""
from .node import Node, Constant
from ..utils.shapes import broadcasted_shape


class Add(Node):

    def __init__(self, *nodes, name=None):
        if len(nodes) < 2:
            raise ValueError("Give at least two parents")
        from .gaussian_markov_chain import GaussianMarkovChain
        nodes = [n.as_gaussian() if isinstance(n, GaussianMarkovChain) else n for n in nodes]
        super().__init__(*nodes, plates=(), dims=((), ()), name=name)
        shape = None
        for p in self.parents:
            if not isinstance(p, Constant):
                if len(p.dims) != 2:
                    raise ValueError('Add needs parents with Gaussian moments; %s has none'
                                     % p.name)
                if shape is not None and tuple(p.dims[0]) != shape:
                    raise ValueError("Nodes do not have identical shapes")
                shape = tuple(p.dims[0])
        if shape is None:
            raise ValueError('Add needs at least one node parent')
        nd = len(shape)
        plates = ()
        for p in self.parents:
            if isinstance(p, Constant):
                full = p.value.shape
                if nd and tuple(full[len(full) - nd:]) != shape:
                    raise ValueError("Nodes do not have identical shapes")
                plates = broadcasted_shape(plates, full[:len(full) - nd] if nd else full)
            else:
                plates = broadcasted_shape(plates, p.plates)
        self.shape = shape
        self.ndim = nd
        self.dims = (shape, shape + shape)
        self.plates = plates


class ConcatGaussian(Node):
    """``ConcatGaussian(X1, X2, ...)``: Gaussian vectors stacked into one vector along the
    variable axis (reference concat_gaussian.py:15-116).  The parents are independent under
    q: the cross blocks of <xx^T> are products of the means."""

    def __init__(self, *nodes, name=None):
        from .gaussian_markov_chain import GaussianMarkovChain
        nodes = [n.as_gaussian() if isinstance(n, GaussianMarkovChain) else n for n in nodes]
        super().__init__(*nodes, plates=(), dims=((), ()), name=name)
        sizes, plates = [], ()
        for p in self.parents:
            if isinstance(p, Constant):
                if p.value.ndim < 1:
                    raise ValueError("Input nodes must be (Gaussian) vectors")
                sizes.append(p.value.shape[-1])
                plates = broadcasted_shape(plates, p.value.shape[:-1])
            else:
                if len(p.dims) != 2 or len(p.dims[0]) != 1:
                    raise ValueError("Input nodes must be (Gaussian) vectors")
                sizes.append(p.dims[0][0])
                plates = broadcasted_shape(plates, p.plates)
        self.sizes = [int(s) for s in sizes]
        D = int(sum(sizes))
        self.offsets = [0]
        for sz in self.sizes:
            self.offsets.append(self.offsets[-1] + sz)
        self.shape = (D,)
        self.ndim = 1
        self.dims = ((D,), (D, D))
        self.plates = plates
