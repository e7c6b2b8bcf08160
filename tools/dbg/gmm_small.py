import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from oracle.gmm import GMMOracle, make_gmm_data
from test_gmm_gpu import _build
for (N, D, K) in [(17, 3, 4), (100, 4, 16), (40, 4, 16), (100, 4, 15), (100, 3, 16), (16, 4, 16), (32, 4, 16)]:
    y, lab0 = make_gmm_data(N, D, K, seed=N + D + K)
    Q = _build(y, lab0, K)
    o = GMMOracle(y, lab0, K)
    try:
        Q.update(Q['mu'], Q['Lambda'], Q['z'], verbose=False)
    except Exception as e:
        print(N, D, K, 'EXC', e)
    o.iterate(1)
    r = Q['z'].u[0]
    print(N, D, K, 'max|r - oracle|', np.abs(r - o.r).max(), 'rowsum err', np.abs(r.sum(1) - 1).max())
    bad = np.argwhere(np.abs(r - o.r) > 1e-9)
    print('  bad entries', len(bad), bad[:10].tolist())
    if len(bad):
        n, k = bad[0]
        print('  r', r[n], '\n  o', o.r[n])
