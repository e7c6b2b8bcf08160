#!/usr/bin/env python
"""Linear state-space model over a batch of sequences -- the model of bayespy/demos/lssm.py:34-103
with a sequence plate (BASELINE.json config 5) on ``bayespy_amd``.  All sequences share dynamics and
noise, so the fused block computes ONE covariance recursion over time and per-sequence mean
recursions; (B, T, D, D) arrays never exist."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply, GaussianMarkovChain  # noqa: E402
from bayespy_amd.inference import VB                                               # noqa: E402
from bayespy_amd.inference import transformations                                  # noqa: E402

np.random.seed(3)
M, B, T, D = 6, 2000, 200, 3
a_true = 0.95 * np.linalg.qr(np.random.randn(D, D))[0]
c_true = np.random.randn(M, D)
x = np.zeros((B, T, D))
x[:, 0] = np.random.randn(B, D)
for t in range(1, T):
    x[:, t] = x[:, t - 1] @ a_true.T + np.random.randn(B, D)
y = np.einsum('md,btd->mbt', c_true, x) + 0.5 * np.random.randn(M, B, T)

alpha = Gamma(1e-5, 1e-5, plates=(D,), name='alpha')
A = GaussianARD(0, alpha, shape=(D,), plates=(D,), name='A')
A.initialize_from_value(np.identity(D))
X = GaussianMarkovChain(np.zeros(D), 1e-3 * np.identity(D), A, np.ones(D), n=T, plates=(B,),
                        name='X')
gamma = Gamma(1e-5, 1e-5, plates=(D,), name='gamma')
C = GaussianARD(0, gamma, shape=(D,), plates=(M, 1, 1), name='C')
C.initialize_from_random()
tau = Gamma(1e-5, 1e-5, name='tau')
F = SumMultiply('i,i', C, X, name='F')
Y = GaussianARD(F, tau, name='Y')
Y.observe(y)

Q = VB(X, C, gamma, A, alpha, tau, F, Y)        # update order of demos/lssm.py:103: X first
print('engine:', type(Q.plans[0]).__name__)
# the rotation speed-up of demos/lssm.py:134-190: the state space (X, A, C) is rotated after every
# iteration; the D x D optimisation runs on the host on plate sums, the means rotate on the device
rotX = transformations.RotateGaussianMarkovChain(X, transformations.RotateGaussianARD(A, alpha))
rot = transformations.RotationOptimizer(rotX, transformations.RotateGaussianARD(C, gamma), D)
Q.set_callback(rot.rotate)
Q.update(repeat=30, tol=1e-7)
print('noise sd: true 0.5, estimated %.3f' % (1.0 / np.sqrt(tau.u[0])))
ev = np.sort(np.abs(np.linalg.eigvals(A.u[0])))[::-1]
print('|eigenvalues| of <A>: %s (true dynamics: 0.95 on all)' % np.round(ev, 3))
