import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import bayespy_amd.nodes as nodes
from bayespy_amd.inference import VB
from models import build_pca
from oracle.pca import make_pca_data
y, x0 = make_pca_data(100000, 128, 32, seed=1)
Q = build_pca(nodes, VB, y, x0, 32)
Q.update(repeat=3, verbose=False)
plan = Q.plans[0]
L = plan.layout
print('cycles build/gj/products/final', plan.state[L.off_scal+4:L.off_scal+8].cpu().numpy())
