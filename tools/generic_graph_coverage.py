"""Which of the test-suite models record a sweep graph, and why the others do not."""
import os, sys, collections, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import bayespy_amd.nodes as nodes
from bayespy_amd.inference import VB
import bayespy_amd.inference.plans.graph_iter as GI
import models
stats = collections.Counter()
orig = GI.GraphIteration._graph_record
def rec(self, upd, bound, key):
    r = orig(self, upd, bound, key)
    stats['recorded' if r is not None else 'declined: ' + str(self._g_disabled)] += 1
    return r
GI.GraphIteration._graph_record = rec
G = os.path.join(R, 'tests', 'golden')
def load(name):
    f = np.load(os.path.join(G, name)); return {k[3:]: f[k] for k in f.files if k.startswith('in_')}
warnings.simplefilter('ignore')
for fn, gold in ((models.run_count_node_cases, 'count_nodes.npz'), (models.run_plate_node_cases, 'plate_nodes.npz'),
                 (models.run_markov_chain_cases, 'markov_chains.npz'), (models.run_slice_cases, 'slice_nodes.npz'),
                 (models.run_switching_case, 'switching_lssm.npz'), (models.run_varying_case, 'varying_lssm.npz'),
                 (models.run_concat_gaussian_case, 'concat_gaussian.npz'), (models.run_constant_parent_cases, 'constant_parents.npz'),
                 (models.run_gaussian_gamma_cases, 'gaussian_gamma.npz')):
    before = dict(stats)
    try:
        fn(nodes, VB, load(gold), engine='generic') if 'engine' in fn.__code__.co_varnames or True else None
    except TypeError:
        fn(nodes, VB, load(gold))
    except Exception as e:
        print(fn.__name__, 'ERROR', type(e).__name__, str(e)[:100])
    print(fn.__name__, {k: v - before.get(k, 0) for k, v in stats.items() if v - before.get(k, 0)})
