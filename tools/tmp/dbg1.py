import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import bayespy_amd.nodes as nodes
from bayespy_amd.inference import VB
from models import build_pca
g = np.load('/root/repo/tests/golden/pca_n500_d6_k3.npz')
y, x0 = g['y'], g['x0']
Q = build_pca(nodes, VB, y, x0, 3, engine='generic')
plan = Q.plans[0]
F, W, X, Y = Q['F'], Q['W'], Q['X'], Q['Y']
mF = plan._messages_from_children(F)
print('Y->F', [np.shape(m.numpy()) for m in mF], mF[1].numpy())
print(' m0 err', np.abs(mF[0].numpy() - 1.0*y).max())
mW = plan._message_to_parent(F, 0)
print('F->W shapes', [m.shape for m in mW])
print(' m0 err', np.abs(mW[0].numpy()[:,0,:] - y @ x0).max(), ' m1 err', np.abs(mW[1].numpy()[0,0] - (-0.5)*(x0.T@x0)).max())
print(mW[1].numpy()[0,0], -0.5*(x0.T@x0))
