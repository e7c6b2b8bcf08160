"""
SumMultiply / Dot deterministic nodes (reference:
bayespy/inference/vmp/nodes/dot.py:19-291, :636-644).

``SumMultiply('i,i', W, X)`` is an einsum-like product of Gaussian-moment
parents summed over the repeated keys; plates broadcast like NumPy.  The node
itself is metadata: its moments (dot.py:316-415) and its messages to the
parents ("THE BEEF", dot.py:580-581) are computed inside the kernels of the
plan that owns it -- for the PCA block they are never materialised at all.
"""
from .node import Node
from ..utils.shapes import broadcasted_shape


def _parse(args):
    """Normalise both calling conventions of the reference (dot.py:118-160):
    ``SumMultiply('ij,j->i', A, B)`` and ``SumMultiply(A, [0,1], B, [1], [0])``."""
    if len(args) > 0 and isinstance(args[0], str):
        spec = args[0].replace(' ', '')
        nodes = list(args[1:])
        if '->' in spec:
            lhs, out = spec.split('->')
        else:
            lhs, out = spec, ''
        ins = lhs.split(',')
        if len(ins) != len(nodes):
            raise ValueError('Number of parents (%d) does not match the number of key lists '
                             'in %r' % (len(nodes), args[0]))
        letters = sorted(set(''.join(ins) + out))
        code = {c: i for i, c in enumerate(letters)}
        in_keys = [[code[c] for c in s] for s in ins]
        out_keys = [code[c] for c in out]
    else:
        nodes = list(args[0::2])
        keys = list(args[1::2])
        if len(args) % 2 == 1:
            nodes, keys, out_keys = list(args[0:-1:2]), list(args[1:-1:2]), list(args[-1])
        else:
            out_keys = []
        in_keys = [list(k) for k in keys]
    for k in out_keys:
        if not any(k in ik for ik in in_keys):
            raise ValueError('Output key %r does not appear in the inputs' % (k,))
    return nodes, in_keys, out_keys


class SumMultiply(Node):

    def __init__(self, *args, name=None, **kwargs):
        nodes, in_keys, out_keys = _parse(args)
        if len(nodes) < 1:
            raise ValueError('SumMultiply needs at least one parent')
        # a Gaussian Markov chain parent is seen through its Gaussian view (moment
        # converter search of the reference, node.py:110-179)
        from .gaussian_markov_chain import GaussianMarkovChain
        from .node import GaussianConstant
        nodes = [n.as_gaussian() if isinstance(n, GaussianMarkovChain) else n for n in nodes]
        # numeric parents: constants with the delta moments of a Gaussian (dot.py:186-197)
        nodes = [n if isinstance(n, Node) else GaussianConstant(n, len(k))
                 for n, k in zip(nodes, in_keys)]
        super().__init__(*nodes, plates=(), dims=((), ()), name=name)
        self.in_keys = in_keys
        self.out_keys = out_keys
        size = {}
        plates = ()
        for node, keys in zip(self.parents, in_keys):
            shp = node.dims[0] if node.dims and len(node.dims[0]) == len(keys) else None
            if shp is None:
                raise ValueError('Parent %s has %d variable axes but %d keys were given'
                                 % (node.name, len(node.dims[0]) if node.dims else 0,
                                    len(keys)))
            for k, s in zip(keys, shp):
                if size.setdefault(k, s) != s:
                    raise ValueError('Key %r has inconsistent sizes %d and %d'
                                     % (k, size[k], s))
            plates = broadcasted_shape(plates, node.plates)
        self.key_sizes = size
        shape = tuple(size[k] for k in out_keys)
        self.shape = shape
        self.ndim = len(shape)
        self.dims = (shape, shape + shape)
        self.plates = plates


class Dot(SumMultiply):
    """Inner product over the last axis of all parents (dot.py:636-644)."""

    def __init__(self, *nodes, name=None, **kwargs):
        super().__init__(','.join(['i'] * len(nodes)), *nodes, name=name, **kwargs)
