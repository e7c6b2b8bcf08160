#!/usr/bin/env python
"""PCA with values missing at random -- bayespy/demos/pca.py:80-94 of the reference on
``bayespy_amd``: an array mask in ``Y.observe`` (NaN placeholders at the missing entries are
fine), the rotation speed-up as the VB callback, predictions of the missing entries at the end.
The model is the one line of the reference; the fused missing-data block takes it (D <= 128,
K <= 32), other sizes run on the generic engine."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply                      # noqa: E402
from bayespy_amd.inference import VB                                               # noqa: E402
from bayespy_amd.inference.vmp import transformations                              # noqa: E402

np.random.seed(41)
D, N, K, K_true = 30, 20000, 10, 4
w = np.random.randn(D, K_true)
x = np.random.randn(K_true, N)
f = w @ x
y = f + 0.3 * np.random.randn(D, N)
mask = np.random.rand(D, N) < 0.8
y_obs = np.where(mask, y, np.nan)

alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
F = SumMultiply('i,i', W, X, name='F')
tau = Gamma(1e-2, 1e-2, name='tau')
Y = GaussianARD(F, tau, name='Y')
X.initialize_from_random()
W.initialize_from_random()
Y.observe(y_obs, mask=mask)

Q = VB(Y, F, W, X, tau, alpha)
print('engine:', type(Q.plans[0]).__name__)
rot = transformations.RotationOptimizer(transformations.RotateGaussianARD(W, alpha),
                                        transformations.RotateGaussianARD(X), K)
Q.set_callback(rot.rotate)
Q.update(repeat=50, tol=1e-6)

pred = F.u[0]                       # <w_d . x_n> for every entry, observed or not
rmse_missing = np.sqrt(np.mean((pred[~mask] - f[~mask]) ** 2))
print('noise sd: true 0.3, estimated %.3f' % (1.0 / np.sqrt(tau.u[0])))
print('RMSE of the reconstruction at the missing entries: %.3f' % rmse_missing)
print('components in use:', int(np.sum(1.0 / alpha.u[0] > 1e-2)), 'of', K)
