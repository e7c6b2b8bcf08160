"""CPU: pure host logic that needs no device -- plate multipliers (node.py:294-301,
:589-632), sharding declarations, Multinomial validation (multinomial.py:66-82, :141-152),
rotation-block argument checks (transformations.py:376-440), checkpoint container."""
import numpy as np
import pytest

import bayespy_amd.nodes as nodes
from bayespy_amd.utils.shapes import multiplier_shape, multiplier_factor, broadcasting_multiplier


def test_multiplier_algebra():
    assert multiplier_shape(None) == ()
    assert multiplier_shape((10.0,), ()) == (10.0,)
    assert multiplier_shape(None, (4,), (1, 4)) == (1, 4)
    assert multiplier_shape((2, 1), (3,)) == (2, 3)
    with pytest.raises(ValueError):
        multiplier_shape((2,), (3,))
    # the factor a child applies to a message: its multiplier where the parent has none
    assert multiplier_factor((10.0,), ()) == 10.0
    assert multiplier_factor((10.0,), (10.0,)) == 1.0
    assert multiplier_factor((2, 5), (1, 5)) == 2.0
    assert multiplier_factor((), ()) == 1.0
    assert broadcasting_multiplier((3, 4), (1, 4), (4,)) == 3


def test_nodes_inherit_multipliers_and_partitions():
    alpha = nodes.Dirichlet(np.ones(3), name='alpha')
    z = nodes.Categorical(alpha, plates=(5,), plates_multiplier=(12.0,), name='z')
    mu = nodes.GaussianARD(0, 1, shape=(2,), plates=(3,), name='mu')
    y = nodes.Mixture(z, nodes.Gaussian, mu, np.identity(2), name='y')
    assert alpha.plates_multiplier == () and mu.plates_multiplier == ()
    assert z.plates_multiplier == (12.0,) and y.plates_multiplier == (12.0,)
    z.plates_multiplier = (3.0,)
    assert y.plates_multiplier == (3.0,)
    x = nodes.GaussianARD(0, 1, shape=(2,), plates=(1, 7), name='x')
    assert x.shard(-1) is x and x._shard_axis == -1
    assert x.shard(1)._shard_axis == -1
    with pytest.raises(ValueError):
        x.shard(2)
    with pytest.raises(ValueError):
        x.shard(1.5)


def test_multinomial_validation():
    p = nodes.Dirichlet(np.array([1.0, 2.0, 3.0]))
    m = nodes.Multinomial(np.array([4, 2, 7]), p)
    assert m.plates == (3,) and m.dims == ((3,),)
    assert nodes.Multinomial(5, p, plates=(2, 6)).plates == (2, 6)
    with pytest.raises(ValueError, match='integer'):
        nodes.Multinomial(2.5, p)
    with pytest.raises(ValueError, match='non-negative'):
        nodes.Multinomial(-1, p)
    with pytest.raises(ValueError, match='sum to the number of trials'):
        m.observe(np.array([[4, 0, 0], [1, 1, 1], [7, 0, 0]]))
    with pytest.raises(ValueError, match='non-negative'):
        m.observe(np.array([[5, -1, 0], [2, 0, 0], [7, 0, 0]]))
    with pytest.raises(ValueError):
        m.observe(np.array([[4, 0], [2, 0], [7, 0]]))


def test_rotation_block_argument_checks():
    from bayespy_amd.inference import transformations as T
    alpha = nodes.Gamma(1e-2, 1e-2, plates=(4,))
    W = nodes.GaussianARD(0, alpha, shape=(4,), plates=(6, 1))
    X = nodes.GaussianARD(0, 1, shape=(4,), plates=(1, 9))
    assert T.RotateGaussianARD(W, alpha).nodes() == [W, alpha]
    assert T.RotateGaussianARD(X).nodes() == [X]
    with pytest.raises(ValueError, match='Too many'):
        T.RotateGaussianARD(W, alpha, alpha)
    with pytest.raises(ValueError):
        T.RotateGaussianARD(W, nodes.Gamma(1e-2, 1e-2, plates=(4,)))     # not W's precision
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(nodes.GaussianARD(1.0, 1, shape=(4,), plates=(3,)))   # non-zero mean
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(nodes.GaussianARD(0, 1, plates=(3,)))                # scalar node
    with pytest.raises(NotImplementedError):
        T.RotateGaussianARD(X).setup(plate_axis=0)


def test_checkpoint_container_round_trip(tmp_path):
    from bayespy_amd.inference.checkpoint import Writer, Reader
    fn = str(tmp_path / 'c.bin')
    w = Writer(fn)
    w.put('nodes/a/u0', np.arange(6.0).reshape(2, 3))
    w.put('iter', 7)
    w.put('converged', False)
    w.close()
    r = Reader(fn)
    assert r.has('nodes/a/u0') and not r.has('nodes/b/u0')
    np.testing.assert_array_equal(r.get('nodes/a/u0'), np.arange(6.0).reshape(2, 3))
    assert int(r.get('iter')) == 7 and not bool(r.get('converged'))
    with pytest.raises(KeyError):
        r.get('missing')
    r.close()
