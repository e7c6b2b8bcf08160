import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from oracle.gmm import GMMOracle, make_gmm_data
from test_gmm_gpu import _build
np.set_printoptions(linewidth=200, precision=4)
for (N, D, K) in [(16, 4, 16), (16, 3, 16)]:
    y, lab0 = make_gmm_data(N, D, K, seed=N + D + K)
    Q = _build(y, lab0, K)
    Q.update(Q['mu'], Q['Lambda'], verbose=False)
    p = Q.plans[0]
    L = p.layout
    p.kernels.prepare_z(D, K, False, p.state)
    st = p.state.cpu().numpy()
    F2P = int(L.F2P); KP = int(L.KP)
    C = st[L.off_C:L.off_C + KP * F2P].reshape(KP, F2P)
    print(N, D, K, 'C finite', np.isfinite(C[:K]).all(), 'F2P', F2P, 'DP', L.DP)
    # host features in compact order
    feats = []
    for a in range(D):
        for b in range(a, D):
            feats.append(y[:, a] * y[:, b])
    for d in range(D):
        feats.append(y[:, d])
    feats.append(np.ones(N))
    F = np.stack(feats, 1)
    phi = F @ C[:K, :F.shape[1]].T
    m = phi.max(1, keepdims=True)
    r_ref = np.exp(phi - m); r_ref /= r_ref.sum(1, keepdims=True)
    p.kernels.pass_(p.Yd, N, D, K, p.Rd, p.state, p.ws)
    r = p.Rd.cpu().numpy().reshape(N, K)
    print(' r finite', np.isfinite(r).all(), 'max diff', np.nanmax(np.abs(r - r_ref)))
    print(' nan rows', np.argwhere(~np.isfinite(r).all(1)).ravel().tolist(), 'nan cols', np.argwhere(~np.isfinite(r).all(0)).ravel().tolist())
    print(r[0])
    print(r_ref[0])
