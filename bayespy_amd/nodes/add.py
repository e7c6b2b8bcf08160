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
