"""
GPU: a node type defined OUTSIDE the package through the extension hook
(bayespy_amd.inference.register_family, plans/extension.py) -- the reference's documented contract
for new nodes (doc/source/dev_guide/writingnodes.rst; stochastic.py:16-80, expfamily.py:17-70): a
Distribution class with the five formulas under the reference's names and a node class that fixes
plates and dims.  The node is the reference's Poisson (poisson.py:25-177) written again from its
documentation, registered, and run through the live-reference golden cases of the built-in one
(tests/golden/count_nodes.npz: Poisson counts with Gamma rates, a Poisson mixture).
"""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _user_poisson():
    from bayespy_amd import darray as da
    from bayespy_amd.nodes import Stochastic, Constant, Gamma
    from bayespy_amd.utils.shapes import broadcasted_shape

    class PoissonDistribution:
        """poisson.py:52-120 under the reference's method names; plain arithmetic + darray functions."""
        finite_phi = True

        def compute_message_to_parent(self, parent, index, u, u_lambda):
            if index == 0:
                return [-1, u[0]]
            raise ValueError('Index out of bounds')

        def compute_phi_from_parents(self, u_lambda, mask=True):
            return [u_lambda[1]]

        def compute_moments_and_cgf(self, phi, mask=True):
            u0 = da.exp(phi[0])
            return [u0], -u0

        def compute_cgf_from_parents(self, u_lambda):
            return -u_lambda[0]

        def compute_fixed_moments_and_f(self, x, mask=True):
            return [x], -da.gammaln(x + 1)

        def compute_fixed_parent_moments(self, index, x):
            return [x, da.log(x)]                       # GammaMoments of a numeric rate, gamma.py:60-75

        def random(self, phi0, plates=None):
            return np.random.poisson(np.exp(np.broadcast_to(phi0, plates)))

    class UserPoisson(Stochastic):
        _parent_count = 1

        def __init__(self, l, plates=None, name=None, plates_multiplier=None):
            super().__init__(l, plates=(), dims=((),), name=name)
            self._plates_multiplier_arg = plates_multiplier
            par = self.parents[0]
            pplates = par.value.shape if isinstance(par, Constant) else par.plates
            given = tuple(plates) if plates is not None else ()
            self.plates = broadcasted_shape(given, pplates)

        def _check_value_shape(self, x):
            x = np.asarray(x)
            if np.any(x != np.round(x)) or np.any(x < 0):
                raise ValueError('Values must be non-negative integers')

    return UserPoisson, PoissonDistribution


def test_user_defined_node_through_the_extension_hook(golden_dir):
    import bayespy_amd.nodes as N_
    from bayespy_amd.inference import VB, register_family, unregister_family
    from bayespy_amd.inference.plans.extension import DistributionAdapter
    from models import run_count_node_cases
    UserPoisson, PoissonDistribution = _user_poisson()
    with pytest.raises(NotImplementedError, match='register_family'):
        lam = N_.Gamma(1.0, 1.0)
        x = UserPoisson(lam, plates=(3,))
        VB(x, lam, engine='generic')
    register_family(UserPoisson, PoissonDistribution)
    try:
        ns = types.SimpleNamespace(**{k: getattr(N_, k) for k in N_.__all__})
        ns.Poisson = UserPoisson
        f = np.load(os.path.join(golden_dir, 'count_nodes.npz'))
        g = {k[3:]: f[k] for k in f.files if k.startswith('in_')}
        res = run_count_node_cases(ns, VB, g)
        seen = 0
        for k, v in res.items():
            if not (k.startswith('poi_') or k.startswith('pmm_')) or k not in f.files:
                continue
            seen += 1
            if isinstance(v, list):
                for i, a in enumerate(v):
                    np.testing.assert_allclose(a, f[k][i] if f[k].dtype == object else f[k][i],
                                               rtol=1e-8, atol=1e-9, err_msg='%s[%d]' % (k, i))
            else:
                np.testing.assert_allclose(v, f[k], rtol=1e-8, atol=1e-8, err_msg=k)
        assert seen >= 4
        # the node really ran on the adapter (not on the built-in Poisson family)
        lam = N_.Gamma(2.0, 1.0, name='lam')
        x = UserPoisson(lam, plates=(5,), name='x')
        x.observe(np.array([1, 0, 3, 2, 2]))
        Q = VB(x, lam, engine='generic')
        assert isinstance(Q.plans[0].family[id(x)], DistributionAdapter)
        Q.update(repeat=3, verbose=False)
        # conjugate answer: Gamma(2 + sum x, 1 + 5)
        np.testing.assert_allclose(np.asarray(lam.u[0]), (2.0 + 8) / (1.0 + 5), rtol=1e-12)
        with pytest.raises(ValueError):
            x.observe(np.array([0.5, 1, 1, 1, 1]))
        with pytest.raises(TypeError, match='missing'):
            register_family(UserPoisson, object)
    finally:
        unregister_family(UserPoisson)
