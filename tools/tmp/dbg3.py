import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from test_chain_gpu import _build_lssm
g = np.load('/root/repo/tests/golden/lssm.npz')
r = np.load('/root/repo/tools/tmp/step_ref.npz')
Q, track = _build_lssm(g, 'lssm1g', None, True)
Q['C'].update(); Q['gamma'].update(); Q['X'].update()
X = track['X']; plan = Q.plans[0]
st = plan.state[id(X)]
print('nu u', track['nu'].u, r['nu_u0'], r['nu_u1'])
for i in range(3):
    a = st.phi[i].numpy(); b = r['X_phi%d' % i]
    print('X phi', i, a.shape, b.shape, np.abs(np.broadcast_to(a, b.shape) - b).max())
for i in range(3):
    a = X.u[i]; b = r['X_u%d' % i]
    print('X u', i, np.abs(a - b).max())
Q['A'].update()
A = track['A']; sa = plan.state[id(A)]
for i in range(2):
    a = sa.phi[i].numpy(); b = r['A_phi%d' % i]
    print('A phi', i, a.shape, b.shape, np.abs(np.broadcast_to(a, b.shape) - b).max())
    print(a if i == 0 else '', b if i == 0 else '')
