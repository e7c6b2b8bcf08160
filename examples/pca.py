#!/usr/bin/env python
"""Probabilistic PCA with ARD and the rotation speed-up -- doc/source/examples/pca.rst of the
reference with ``bayespy`` replaced by ``bayespy_amd`` (needs an MI355X)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply          # noqa: E402
from bayespy_amd.inference import VB                                   # noqa: E402
from bayespy_amd.inference.transformations import RotateGaussianARD, RotationOptimizer   # noqa: E402

np.random.seed(1)
M, N, D = 20, 100, 10
y = np.random.randn(M, 2) @ np.random.randn(2, N) + 0.1 * np.random.randn(M, N)

X = GaussianARD(0, 1, plates=(1, N), shape=(D,), name='X')
alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
C = GaussianARD(0, alpha, plates=(M, 1), shape=(D,), name='C')
F = SumMultiply('d,d->', X, C, name='F')
tau = Gamma(1e-5, 1e-5, name='tau')
Y = GaussianARD(F, tau, name='Y')
Y.observe(y)

Q = VB(Y, X, C, alpha, tau)
C.initialize_from_random()
R = RotationOptimizer(RotateGaussianARD(X), RotateGaussianARD(C, alpha), D)
Q.set_callback(R.rotate)
Q.update(repeat=1000)
print('effective dimensionality:', int(np.sum(np.ravel(alpha.u[0]) < 100)))
