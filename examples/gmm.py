#!/usr/bin/env python
"""Gaussian mixture model -- bayespy/demos/mog.py of the reference on ``bayespy_amd``."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import Dirichlet, Categorical, Gaussian, Wishart, Mixture   # noqa: E402
from bayespy_amd.inference import VB                                               # noqa: E402

np.random.seed(2)
N, D, K = 2000, 2, 8
centers = 4 * np.random.randn(3, D)
y = centers[np.random.randint(3, size=N)] + 0.5 * np.random.randn(N, D)

alpha = Dirichlet(1e-3 * np.ones(K), name='alpha')
Z = Categorical(alpha, plates=(N,), name='z')
mu = Gaussian(np.zeros(D), 1e-3 * np.identity(D), plates=(K,), name='mu')
Lambda = Wishart(D, 1e-2 * np.identity(D), plates=(K,), name='Lambda')
Y = Mixture(Z, Gaussian, mu, Lambda, name='Y')
Z.initialize_from_random()
Y.observe(y)

Q = VB(Y, mu, Lambda, Z, alpha)
Q.update(repeat=200)
counts = np.sum(Z.u[0], axis=0)
print('clusters in use:', int(np.sum(counts > 1)), 'of', K)
