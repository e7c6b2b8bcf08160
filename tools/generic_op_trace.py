"""Op-level trace of one generic-engine iteration (argument: pca = BASELINE config 2, gmm = the
mixture of bench.py --config generic_gmm): every fused elementwise
launch and every sum_multiply / GEMM launch with shapes and its synchronous duration."""
import os, sys, time, traceback
# launch by launch: no replay from the sweep graph, no queue of small operations
os.environ.setdefault('BAYESPY_AMD_GRAPH', '0')
os.environ.setdefault('BAYESPY_AMD_SMALL_QUEUE', '0')   # (the trace wants every operation as a launch)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayespy_amd import darray
from bayespy_amd.utils import misc
import bayespy_amd.inference.plans.generic as G
LOG = []
ON = [False]
def caller():
    for fs in reversed(traceback.extract_stack()[:-3]):
        if '/plans/' in fs.filename or 'utils/linalg.py' in fs.filename or 'utils/misc.py' in fs.filename:
            return '%s:%d %s' % (os.path.basename(fs.filename), fs.lineno, fs.name)
    return '?'
def wrap(mod, name, kind):
    orig = getattr(mod, name)
    def f(*a, **k):
        if not ON[0]:
            return orig(*a, **k)
        torch.cuda.synchronize(); t = time.perf_counter()
        ON[0] = False
        try:
            r = orig(*a, **k)
        finally:
            ON[0] = True
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        shapes = [tuple(x.shape) for x in a if hasattr(x, 'shape')]
        if kind == 'sm':
            shapes = [tuple(x.shape) for x in a[0]] + ['->', tuple(a[3])]
        LOG.append((dt * 1e3, kind, str(shapes)[:110], caller()))
        return r
    setattr(mod, name, f)
wrap(misc, '_launch_sum_multiply', 'sm')
orig_fuse = darray.fuse
def fuse_t(fn, *ops):
    if not ON[0]:
        return orig_fuse(fn, *ops)
    torch.cuda.synchronize(); t = time.perf_counter()
    r = orig_fuse(fn, *ops)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    LOG.append((dt * 1e3, 'ew', str([tuple(x.shape) for x in ops if hasattr(x, 'shape')])[:110], caller()))
    return r
import bayespy_amd.utils.linalg as LA
for _m in list(sys.modules.values()):
    # (every module of the package that imported `fuse` by name: plans/lazy.py, plans/families/*, ...)
    if getattr(_m, '__name__', '').startswith('bayespy_amd') and getattr(_m, 'fuse', None) is orig_fuse:
        _m.fuse = fuse_t
MODEL = sys.argv[1] if len(sys.argv) > 1 else 'pca'
from bayespy_amd.inference import VB
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(42)
if MODEL == 'pca':
    from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
    N, D, K = 1_000_000, 64, 16
    w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
    x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
    y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
    x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
    alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha'); W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
    X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X'); F = SumMultiply('i,i', W, X, name='F')
    tau = Gamma(1e-2, 1e-2, name='tau'); Y = GaussianARD(F, tau, name='Y')
    X.initialize_from_value(x0[None]); Y.observe(y)
    Q = VB(Y, F, W, X, tau, alpha, engine='generic'); Q.ignore_bound_checks = True
else:
    # the mixture of bench.py --config generic_gmm
    from bayespy_amd.nodes import GaussianARD, Gaussian, Wishart, Dirichlet, Categorical, Mixture
    N, D, K = 100_000, 16, 32
    centers = 3 * torch.randn(K, D, generator=g, device=dev, dtype=torch.float64)
    lab = torch.randint(0, K, (N,), generator=g, device=dev)
    y = centers[lab] + 0.5 * torch.randn(N, D, generator=g, device=dev, dtype=torch.float64)
    lab0 = torch.randint(0, K, (N,), generator=g, device=dev)
    alpha = Dirichlet(1e-3 * np.ones(K), name='alpha'); z = Categorical(alpha, plates=(N,), name='z')
    mu = GaussianARD(0, 1e-3, shape=(D,), plates=(K,), name='mu')
    Lam = Wishart(D, 0.01 * np.identity(D), plates=(K,), name='Lambda')
    Y = Mixture(z, Gaussian, mu, Lam, plates=(N,), name='Y')
    z.initialize_from_value(lab0.cpu().numpy()); Y.observe(y)
    Q = VB(Y, mu, Lam, z, alpha, engine='generic'); Q.ignore_bound_checks = True
Q.update(repeat=2, verbose=False)
torch.cuda.synchronize()
t = time.perf_counter(); Q.update(repeat=3, verbose=False); torch.cuda.synchronize()
print('untraced ms/iter', (time.perf_counter() - t) / 3 * 1e3)
ON[0] = True
Q.update(repeat=1, verbose=False)
ON[0] = False
tot = sum(l[0] for l in LOG)
print('traced launches', len(LOG), 'sum ms', round(tot, 2))
for l in sorted(LOG, key=lambda l: -l[0])[:40]:
    print('%7.3f %-3s %-110s %s' % l)
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/gen_trace_seq_%s.txt' % MODEL, 'w') as f:
    for l in LOG:
        f.write('%7.3f %-3s %-110s %s\n' % l)
print('peak GB', torch.cuda.max_memory_allocated() / 1e9)
