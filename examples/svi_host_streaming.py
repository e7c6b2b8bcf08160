#!/usr/bin/env python
"""Stochastic variational inference with the data kept in HOST memory (it may exceed HBM):
bayespy/demos/stochastic_inference.py:99-133 of the reference on ``bayespy_amd``.  A worker
thread gathers the next mini-batch into pinned memory and copies it host -> HBM on its own
stream while the current one is in use (``HostBatchStream``)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import GaussianARD, Gaussian, Dirichlet, Categorical, Mixture  # noqa: E402
from bayespy_amd.inference import VB                                                 # noqa: E402
from bayespy_amd.utils.streaming import HostBatchStream                              # noqa: E402

np.random.seed(7)
N, D, K, NB = 2_000_000, 2, 10, 20_000
centers = 5 * np.random.randn(4, D)
data = centers[np.random.randint(4, size=N)] + np.random.randn(N, D)      # stays on the host

mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='means')
alpha = Dirichlet(np.ones(K), name='class probabilities')
Z = Categorical(alpha, plates=(NB,), plates_multiplier=(N / NB,), name='classes')
Y = Mixture(Z, Gaussian, mu, np.identity(D), name='observations')
mu.initialize_from_random()
Q = VB(Y, Z, mu, alpha)
Q.ignore_bound_checks = True

steps = 200
delay, forgetting = 1.0, 0.7
batches = [np.random.choice(N, NB, replace=False) for _ in range(steps)]
for n, (y_dev, _) in enumerate(HostBatchStream(data, batches)):
    Y.observe(y_dev)                      # a device tensor: used in place
    Q.update(Z, verbose=False)            # local step on the mini-batch
    step = (n + delay) ** (-forgetting)
    Q.gradient_step(mu, alpha, scale=step)   # global step along the natural gradient
found = mu.u[0][np.argsort(-alpha.u[0])[:4]]
print('four heaviest cluster means:\n', np.round(found, 2))
print('true centres:\n', np.round(centers, 2))
