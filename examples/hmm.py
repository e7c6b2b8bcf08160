#!/usr/bin/env python
"""Hidden Markov model with Gaussian emissions -- doc/source/examples/hmm.rst of the
reference on ``bayespy_amd``."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import Dirichlet, CategoricalMarkovChain, Gaussian, Mixture   # noqa: E402
from bayespy_amd.inference import VB                                                 # noqa: E402

np.random.seed(1)
mu = np.array([[0, 0], [3, 4], [6, 0]])
K, N, std = 3, 200, 2.0
P = 0.9 * np.identity(K) + 0.05 * (np.ones((K, K)) - np.identity(K))
y = np.zeros((N, 2))
state = np.random.choice(K)
for n in range(N):
    y[n] = std * np.random.randn(2) + mu[state]
    state = np.random.choice(K, p=P[state])

a0 = Dirichlet(1e-3 * np.ones(K), name='a0')
A = Dirichlet(1e-3 * np.ones((K, K)), name='A')
Z = CategoricalMarkovChain(a0, A, states=N, name='Z')
Y = Mixture(Z, Gaussian, mu, std ** (-2) * np.identity(2), name='Y')
Y.observe(y)

Q = VB(Y, Z, A, a0)
Q.update(repeat=1000)
print('estimated transition matrix:\n', np.round(np.exp(A.u[0]) / np.exp(A.u[0]).sum(-1, keepdims=True), 2))
