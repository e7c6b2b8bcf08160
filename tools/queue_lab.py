"""Lab: what the queue of small operations collects in one sweep of the generic engine (BASELINE
config 2, engine='generic') and what forces its flushes.  QUEUE_LAB_SM=1 queues plate sums as well;
VMP_QUEUE_TRACE=1 makes the library name the entry point behind every flush (stderr);
QUEUE_LAB_PY=1 adds the Python frame behind every flush this side asks for."""
import os
import sys
import time
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayespy_amd.nodes import GaussianARD, Gamma, SumMultiply
from bayespy_amd.inference import VB
from bayespy_amd.device import get_runtime, Runtime

N, D, K = 1_000_000, 64, 16
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(42)
w = torch.randn(D, K, generator=g, device=dev, dtype=torch.float64)
x = torch.randn(K, N, generator=g, device=dev, dtype=torch.float64)
y = w @ x + 0.1 * torch.randn(D, N, generator=g, device=dev, dtype=torch.float64)
x0 = torch.randn(N, K, generator=g, device=dev, dtype=torch.float64)
alpha = Gamma(1e-2, 1e-2, plates=(K,), name='alpha')
W = GaussianARD(0, alpha, shape=(K,), plates=(D, 1), name='W')
X = GaussianARD(0, 1, shape=(K,), plates=(1, N), name='X')
F = SumMultiply('i,i', W, X, name='F')
tau = Gamma(1e-2, 1e-2, name='tau')
Y = GaussianARD(F, tau, name='Y')
X.initialize_from_value(x0[None]); Y.observe(y)
Q = VB(Y, F, W, X, tau, alpha, engine='generic'); Q.ignore_bound_checks = True
rt = get_runtime()
rt.queue_begin(); rt.queue_end()      # (the runtime sets its default of small_queue_sm on first use)
rt.set_tune('small_queue_sm', int(os.environ.get('QUEUE_LAB_SM', '1')))
for key in ('small_queue_ew_max', 'small_queue_sm_work', 'small_queue_spd'):
    if os.environ.get(key.upper()):
        rt.set_tune(key, int(os.environ[key.upper()]))
if os.environ.get('QUEUE_LAB_PY'):
    orig = Runtime.flush_small

    def traced(self):
        fs = [f for f in traceback.extract_stack()[:-1] if 'bayespy_amd' in f.filename][-3:]
        sys.stderr.write('[py flush] ' + ' < '.join('%s:%d %s' % (os.path.basename(f.filename),
                                                                 f.lineno, f.name)
                                                    for f in reversed(fs)) + '\n')
        return orig(self)
    Runtime.flush_small = traced
Q.update(repeat=2, verbose=False)
torch.cuda.synchronize()
s0 = rt.queue_stats()
sys.stderr.write('==== one sweep ====\n')
t = time.perf_counter()
Q.update(repeat=1, verbose=False)
torch.cuda.synchronize()
s1 = rt.queue_stats()
print('sweep ms', (time.perf_counter() - t) * 1e3, 'queue launches', s1['launches'] - s0['launches'],
      'records', s1['operations'] - s0['operations'])
t = time.perf_counter(); Q.update(repeat=10, verbose=False); torch.cuda.synchronize()
print('ms/iter over 10', (time.perf_counter() - t) * 100, Q.plans[0].graph_info())
print('L', Q.L[Q.iter - 1])
