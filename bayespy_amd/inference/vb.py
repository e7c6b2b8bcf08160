"""
The VB engine: iteration driver, lower-bound bookkeeping, convergence test.

Host-side mirror of ``bayespy.inference.vmp.vmp.VB`` (vmp.py:52-172,
:180-199, :693-764): same constructor/``update`` signature, same iteration
semantics (update the given nodes in order, then evaluate the full lower
bound, warn if it decreased by more than 1e-6, stop when the relative change
drops below ``tol``), same log line.  The numbers come from the compiled
plans, i.e. from HIP kernels.
"""
import os
import time
import warnings

import numpy as np

from ..nodes.node import Node
from .plans import compile_model


def _unique(nodes):
    seen, out = set(), []
    for n in nodes:
        if id(n) not in seen:
            seen.add(id(n))
            out.append(n)
    return out


def _closure(node):
    """The connected model block around ``node`` (for stand-alone node calls)."""
    seen, stack, out = set(), [node], []
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        out.append(n)
        stack.extend(n.parents)
        stack.extend(c for (c, _) in n.children)
    return out


def compile_for_node(node):
    compile_model(_closure(node))


class VB:

    def __init__(self, *nodes, tol=1e-5, autosave_filename=None, autosave_iterations=0,
                 use_logging=False, user_data=None, callback=None, engine=None):
        for i, n in enumerate(nodes):
            if not isinstance(n, Node):
                raise ValueError("Argument number %d is not a node" % (i + 1))
        # vmp.py:87-99: without a name the autosave goes to a temporary file
        self.autosave_iterations = int(autosave_iterations or 0)
        self.autosave_nodes = None
        self.filename = autosave_filename or None
        self._autosave_tmp = None
        self.autosave_filename = autosave_filename or None
        if use_logging:
            import logging
            self.print = logging.getLogger(__name__).info
        else:
            self.print = print
        self.user_data = user_data
        self.model = _unique(nodes)
        names = [n.name for n in self.model]
        if len(set(names)) != len(names):
            raise Exception("Use unique names for nodes.")
        compile_model(self.model, engine=engine)
        self.ignore_bound_checks = False
        self.annealing_changed = False
        self.iter = 0
        self.converged = False
        self.L = np.array(())
        self.cputime = np.array(())
        self.l = {n: np.array(()) for n in self.model}
        self.callback = callback
        self.callback_output = None
        self.tol = tol

    @property
    def plans(self):
        """The execution plans that currently own the model's nodes."""
        out, seen = [], set()
        for n in self.model:
            p = n._plan
            if p is not None and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    # -- containers ---------------------------------------------------------------
    def __getitem__(self, name):
        if isinstance(name, Node):
            return name
        for n in self.model:
            if n.name == name:
                return n
        raise KeyError(name)

    def use_logging(self, use):
        """Route the iteration messages through ``logging`` instead of ``print`` (vmp.py:111-118;
        the reference tests the module-level name there, so its method always raises NameError --
        this one does what its comment says)."""
        import logging
        self.print = logging.getLogger(__name__).info if use else print

    def get_iteration_by_nodes(self):
        """Per-node lower-bound traces, {node: array over iterations} (vmp.py:233-234)."""
        return self.l

    def set_callback(self, callback):
        self.callback = callback

    def set_autosave(self, filename, iterations=None, nodes=None):
        """vmp.py:121-126."""
        self.autosave_filename = filename
        self.filename = filename
        self.autosave_nodes = nodes
        if iterations is not None:
            self.autosave_iterations = int(iterations)

    def _autosave_target(self):
        if not self.autosave_filename:
            import datetime
            import tempfile
            prefix = 'vb_autosave_%s_' % datetime.datetime.today().strftime('%Y%m%d%H%M%S')
            self._autosave_tmp = tempfile.NamedTemporaryFile(prefix=prefix, suffix='.ckpt')
            self.autosave_filename = self._autosave_tmp.name
        return self.autosave_filename

    @staticmethod
    def load_user_data(filename):
        """The ``user_data`` dictionary stored by ``save`` (vmp.py:295-305)."""
        from .checkpoint import Reader
        r = Reader(filename)
        try:
            out = {k[len('user_data/'):]: np.array(r.get(k)) for k in r.keys()
                   if k.startswith('user_data/')}
            import json
            for k in r.keys():
                if k.startswith('user_data_json/'):
                    out[k[len('user_data_json/'):]] = json.loads(
                        bytes(np.asarray(r.get(k), dtype=np.uint8)).decode('utf-8'))
            return out
        finally:
            r.close()

    def _append_iterations(self, k):
        self.L = np.append(self.L, np.full(k, np.nan))
        self.cputime = np.append(self.cputime, np.full(k, np.nan))
        for n in self.model:
            self.l[n] = np.append(self.l[n], np.full(k, np.nan))

    # -- the loop (vmp.py:132-172) ---------------------------------------------------
    def update(self, *nodes, repeat=1, plot=False, tol=None, verbose=True, tqdm=None):
        if len(nodes) == 0:
            nodes = self.model
        if tqdm is not None:
            tqdm = tqdm(total=repeat)
        i = 0
        try:
            while repeat is None or i < repeat:
                t = time.time()
                if not self._graph_sweep(nodes):
                    for node in nodes:
                        X = self[node]
                        if hasattr(X, 'update') and callable(X.update):
                            X.update()
                cputime = time.time() - t
                i += 1
                if tqdm is not None:
                    tqdm.update()
                if self._end_iteration_step(None, cputime, tol=tol, verbose=verbose):
                    return
        finally:
            # plans may keep plate-sized work in flight on their own streams across
            # iterations; order the caller's stream after it before handing back control
            for p in self.plans:
                fin = getattr(p, 'finish', None)
                if fin is not None:
                    fin()

    def _graph_sweep(self, nodes):
        """A sweep over ``nodes`` and the bound terms of the model as ONE recorded HIP graph, when
        a single generic plan owns the whole model and has seen this sweep twice
        (plans/graph_iter.py).  False: not done, the caller visits the nodes one by one."""
        plan = None
        for n in self.model:
            p = n._plan
            if p is None or (plan is not None and p is not plan):
                return False
            plan = p
        run = getattr(plan, 'graph_iteration', None)
        if run is None:
            return False
        upd = []
        for node in nodes:
            X = self[node]
            if not (hasattr(X, 'update') and callable(X.update)):
                continue
            if getattr(X, '_plan', None) is not plan:
                return False
            if X.observed and getattr(X, '_fully_observed', True):
                continue
            upd.append(X)
        return bool(run(upd, list(self.model)))

    # -- persistence (vmp.py:237-356) ------------------------------------------------------
    def _shard_info(self, nodes):
        """(rank, world) when the model holds a plate sharded over several ranks, else None."""
        try:
            import torch.distributed as dist
        except Exception:       # noqa: BLE001
            return None
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
            return None
        if not any(getattr(n, '_shard_axis', None) is not None for n in self.model):
            return None
        return dist.get_rank(), dist.get_world_size()

    def _checkpoint_name(self, filename, nodes):
        """A model with a sharded plate keeps rank-local state: every rank writes / reads its own
        file ``<filename>.rank<r>of<w>`` (one shared name would be a concurrent-write race and
        would restore another rank's shard)."""
        info = self._shard_info(nodes)
        if info is None:
            return filename, None
        root, ext = os.path.splitext(filename)
        return '%s.rank%dof%d%s' % (root, info[0], info[1], ext), info

    def save(self, *nodes, filename=None):
        """Write the state of ``nodes`` (default: all) and the iteration statistics; device
        state is read back once.  Layout: inference/checkpoint.py."""
        from .checkpoint import Writer
        nodes = [self[n] for n in nodes if n is not None] if nodes else list(self.model)
        filename = filename or self.autosave_filename
        if not filename:
            raise Exception("Filename must be given.")
        names = [n.name for n in nodes]
        if len(set(names)) != len(names) or any(nm == '' for nm in names):
            raise Exception("In order to save nodes, they must have (unique) names.")
        filename, info = self._checkpoint_name(filename, nodes)
        w = Writer(filename)
        if info is not None:
            w.put('shard/rank', info[0])
            w.put('shard/world', info[1])
        seen = []
        for n in nodes:
            p = n._plan
            if p is None or any(p is q for q in seen):
                continue
            seen.append(p)
            p.save_state(lambda path, v: w.put(path, v), [m for m in nodes if m._plan is p],
                         len(seen) - 1)
        w.put('L', self.L)
        w.put('cputime', self.cputime)
        w.put('iter', self.iter)
        w.put('converged', bool(self.converged))
        if self.callback_output is not None:
            w.put('callback_output', self.callback_output)
        for n in nodes:
            w.put('boundterms/' + n.name, self.l[n])
        if self.user_data is not None:
            for key, value in self.user_data.items():
                # numeric / string arrays as arrays; anything else (dicts, None, mixed lists) as
                # JSON text -- an object array would be pickled by np.savez and could not be read
                # back by the reader (allow_pickle=False)
                try:
                    arr = np.asarray(value)
                except Exception:       # noqa: BLE001 -- ragged input
                    arr = None
                if arr is not None and arr.dtype != object:
                    w.put('user_data/%s' % key, arr)
                    continue
                import json
                try:
                    text = json.dumps(value)
                except (TypeError, ValueError):
                    raise TypeError('user_data[%r] is neither an array nor JSON-serialisable; '
                                    'it cannot be stored in a checkpoint' % key)
                w.put('user_data_json/%s' % key, np.frombuffer(text.encode('utf-8'), dtype=np.uint8))
        w.close()

    def load(self, *nodes, filename=None, nodes_only=False):
        from .checkpoint import Reader
        nodes = [self[n] for n in nodes if n is not None] if nodes else list(self.model)
        filename = filename or self.autosave_filename
        if not filename:
            raise Exception("Filename must be given.")
        filename, info = self._checkpoint_name(filename, nodes)
        r = Reader(filename)
        try:
            if info is not None and r.has('shard/rank'):
                got = (int(r.get('shard/rank')), int(r.get('shard/world')))
                if got != info:
                    raise ValueError('checkpoint %s holds the shard of rank %d of %d, this process '
                                     'is rank %d of %d' % ((filename,) + got + info))
            seen = []
            for n in nodes:
                p = n._plan
                if p is None or any(p is q for q in seen):
                    continue
                seen.append(p)
                p.load_state(r, [m for m in nodes if m._plan is p], len(seen) - 1)
            if not nodes_only:
                self.L = np.array(r.get('L'))
                self.cputime = np.array(r.get('cputime'))
                self.iter = int(r.get('iter'))
                self.converged = bool(r.get('converged'))
                for n in nodes:
                    if r.has('boundterms/' + n.name):       # a file of an autosave_nodes subset
                        self.l[n] = np.array(r.get('boundterms/' + n.name))
                if r.has('callback_output'):
                    self.callback_output = np.array(r.get('callback_output'))
        finally:
            r.close()

    def gradient_step(self, *nodes, scale=1.0):
        """Update ``nodes`` by a step of length ``scale`` along the Riemannian gradient of the
        lower bound (vmp.py:432-440): with mini-batch children carrying ``plates_multiplier``
        this is the global step of stochastic variational inference
        (demos/stochastic_inference.py:99-133); ``scale=1`` equals a simultaneous VB update."""
        if len(nodes) == 0:
            nodes = self.model
        nodes = [self[n] for n in nodes]
        by_plan = {}
        for n in nodes:
            p = n._plan
            if p is None or not hasattr(p, 'gradient_step'):
                raise NotImplementedError('gradient steps are built for the generic engine; '
                                          'node %s is owned by %s'
                                          % (n.name, type(p).__name__))
            by_plan.setdefault(id(p), (p, []))[1].append(n)
        for p, ns in by_plan.values():
            p.gradient_step(ns, scale=scale)

    def has_converged(self, tol=None):
        return self.converged

    # -- deterministic annealing (vmp.py:665-677) ------------------------------------------
    def set_annealing(self, annealing):
        """Annealing coefficient from (0, 1]; 1 = standard updates.  The next iteration's
        bound is not compared with the previous one (the objective changed)."""
        if not (0.0 < annealing <= 1.0):
            raise ValueError('annealing must be in (0, 1]')
        for node in self.model:
            if hasattr(node, 'annealing'):
                if annealing != 1.0 and node._plan is not None and \
                        not hasattr(node._plan, 'riemannian_gradient'):
                    raise NotImplementedError(
                        "annealing is not built into the fused %s plan; build the engine with "
                        "VB(..., engine='generic')" % type(node._plan).__name__)
                node.annealing = float(annealing)
        self.annealing_changed = True

    # -- parameter vectors and gradients (vmp.py:401-466) ------------------------------------
    # Vectors are lists (nodes) of lists (parameters) of DEVICE arrays: the conjugate-gradient
    # and pattern-search loops below never move plate-sized data to the host.
    def _param_plan(self, node, method):
        node = self[node]
        plan = node._require_plan()
        fn = getattr(plan, method, None)
        if fn is None:
            raise NotImplementedError(
                "%s: the fused %s plan does not expose variational parameters; build the "
                "engine with VB(..., engine='generic')" % (method, type(plan).__name__))
        return node, fn

    def get_parameters(self, *nodes):
        out = []
        for n in nodes:
            node, fn = self._param_plan(n, 'natural_parameters')
            out.append(list(fn(node)))
        return out

    def set_parameters(self, x, *nodes):
        for n, xi in zip(nodes, x):
            node, fn = self._param_plan(n, 'set_parameters')
            fn(node, xi)

    def get_gradients(self, *nodes, euclidian=False):
        rg = []
        for n in nodes:
            node, fn = self._param_plan(n, 'riemannian_gradient')
            rg.append(fn(node))
        if not euclidian:
            return rg
        g = []
        for n, r in zip(nodes, rg):
            node, fn = self._param_plan(n, 'gradient')
            g.append(fn(node, r))
        return rg, g

    @staticmethod
    def dot(x1, x2):
        """Inner product of two parameter vectors: every pair is one device contraction, the
        partial results are added on the device and read once."""
        from ..utils import misc
        from ..darray import asdarray, fuse
        tot = None
        for y1, y2 in zip(x1, x2):
            for z1, z2 in zip(y1, y2):
                z1, z2 = asdarray(z1), asdarray(z2)
                nd = max(z1.ndim, z2.ndim)
                v = misc.sum_multiply(z1, z2, axis=tuple(range(nd))) if nd else \
                    fuse(lambda a, b: a * b, z1, z2)
                tot = v if tot is None else fuse(lambda a, b: a + b, tot, v)
        return 0.0 if tot is None else float(tot.item())

    @staticmethod
    def add(x1, x2, scale=1):
        from ..darray import asdarray, fuse
        s = float(scale)
        return [[fuse(lambda a, b, s_=s: a + s_ * b, asdarray(z1), asdarray(z2))
                 for z1, z2 in zip(y1, y2)] for y1, y2 in zip(x1, x2)]

    # -- gradient-based optimisation (vmp.py:469-660) ------------------------------------------
    def _try_step(self, p, direction, scale, nodes, collapsed):
        """Move ``nodes`` to p + scale * direction and update the collapsed nodes.  Returns
        ('ok', p_new), or ('invalid', None) when a distribution left its domain, or
        ('lowered', None) when the bound dropped; the collapsed nodes are restored then."""
        p_new = self.add(p, direction, scale=scale)
        try:
            self.set_parameters(p_new, *nodes)
        except Exception:
            return 'invalid', None
        saved = self.get_parameters(*collapsed)
        try:
            for node in collapsed:
                self[node].update()
        except Exception:
            self.set_parameters(saved, *collapsed)
            return 'invalid', None
        L = self.compute_lowerbound()
        if self.iter > 0:
            L0 = self.L[self.iter - 1]
            lowered = L < L0 and not np.allclose(L, L0, rtol=1e-8)
        else:
            lowered = False
        if np.isnan(L) or lowered:
            self.set_parameters(saved, *collapsed)
            return 'lowered', None
        return 'ok', p_new

    def optimize(self, *nodes, maxiter=10, verbose=True, method='fletcher-reeves',
                 riemannian=True, collapsed=None, tol=None):
        """Riemannian conjugate-gradient ascent on the variational parameters of ``nodes``
        with the ``collapsed`` nodes kept at their optimum (vmp.py:469-601).  The step length
        grows by sqrt(2) after an accepted step and halves after a rejected gradient step; a
        conjugate direction is dropped for the plain gradient when a step along it leaves the
        domain, or lowers the bound with the step length already below 2^-10."""
        method = method.lower()
        if method not in ('gradient', 'fletcher-reeves'):
            raise Exception("Unknown optimization method: %s" % (method))
        collapsed = list(collapsed) if collapsed is not None else []
        scale = 1.0
        p = self.get_parameters(*nodes)
        norm_prev = 0
        s = None
        for _ in range(maxiter):
            t = time.time()
            if riemannian and method == 'gradient':
                steepest = weight = self.get_gradients(*nodes)
            else:
                rg, g = self.get_gradients(*nodes, euclidian=True)
                weight = g
                steepest = rg if riemannian else g
            beta = 0
            if method == 'fletcher-reeves':
                norm = self.dot(weight, steepest)
                if norm_prev != 0:
                    beta = norm / norm_prev
                norm_prev = norm
            s = self.add(steepest, s, scale=beta) if beta else steepest
            while True:
                status, p_new = self._try_step(p, s, scale, nodes, collapsed)
                if status == 'ok':
                    break
                plain = s is steepest
                if status == 'invalid':
                    if verbose:
                        self.print("CG update was unsuccessful, using gradient and resetting CG")
                    if plain:
                        scale = scale / 2
                    norm_prev = 0
                    s = steepest
                elif plain or scale >= 2 ** (-10):
                    if verbose:
                        self.print("Step decreased the lower bound, halfing step length")
                    scale = scale / 2
                else:
                    if verbose:
                        self.print("CG decreased the lower bound, reset CG.")
                    norm_prev = 0
                    s = steepest
            scale = scale * np.sqrt(2)
            p = p_new
            cputime = time.time() - t
            if self._end_iteration_step('OPT', cputime, tol=tol, verbose=verbose):
                break

    def pattern_search(self, *nodes, collapsed=None, maxiter=3):
        """Pattern search (Honkela et al. 2003; vmp.py:603-662): extrapolate along the change
        of the parameters made by one VB update, with the step length chosen by a scalar
        minimisation of the negative lower bound."""
        from scipy import optimize as sp_optimize
        collapsed = list(collapsed) if collapsed is not None else []
        t = time.time()
        for x in nodes:
            self[x].update()
        for x in collapsed:
            self[x].update()
        p0 = self.get_parameters(*nodes)
        for x in nodes:
            self[x].update()
        p1 = self.get_parameters(*nodes)
        dp = self.add(p1, p0, scale=-1)

        def cost(alpha):
            try:
                self.set_parameters(self.add(p1, dp, scale=alpha), *nodes)
            except Exception:
                return np.inf
            for x in collapsed:
                self[x].update()
            return -self.compute_lowerbound()

        res = sp_optimize.minimize_scalar(cost, bracket=[0, 3], options={'maxiter': maxiter})
        self.set_parameters(self.add(p1, dp, scale=res.x), *nodes)
        for x in collapsed:
            self[x].update()
        self._end_iteration_step('PS', time.time() - t)

    # -- lower bound (vmp.py:180-199) ----------------------------------------------------
    def compute_lowerbound(self, ignore_masked=True):
        if ignore_masked:
            return sum(n.lower_bound_contribution() for n in self.model)
        return sum(n.lower_bound_contribution(ignore_masked=False) for n in self.model)

    def compute_lowerbound_terms(self, *nodes):
        if len(nodes) == 0:
            nodes = self.model
        return {n: n.lower_bound_contribution() for n in nodes}

    def loglikelihood_lowerbound(self):
        # nodes of one plan are asked together so that a plan can answer with a single
        # device -> host read; the sum runs in model order like vmp.py:192-199
        terms = {}
        groups = {}
        for n in self.model:
            p = n._plan
            if p is not None and hasattr(p, 'lower_bound_contributions'):
                groups.setdefault(id(p), (p, []))[1].append(n)
        for p, ns in groups.values():
            for n, v in zip(ns, p.lower_bound_contributions(ns)):
                terms[n] = v
        L = 0.0
        for n in self.model:
            lp = terms[n] if n in terms else n.lower_bound_contribution()
            L += lp
            self.l[n][self.iter] = lp
        return L

    def _end_iteration_step(self, method, cputime, tol=None, verbose=True):
        if self.iter >= len(self.L):
            self._append_iterations(100)
        if callable(self.callback):
            z = self.callback()
            if z is not None:
                z = np.array(z)[..., np.newaxis]
                self.callback_output = z if self.callback_output is None else \
                    np.concatenate((self.callback_output, z), axis=-1)
        t = time.time()
        L = self.loglikelihood_lowerbound()      # device -> host sync point
        cputime += time.time() - t
        self.cputime[self.iter] = cputime
        self.L[self.iter] = L
        if verbose:
            if method:
                self.print("Iteration %d (%s): loglike=%e (%.3f seconds)"
                           % (self.iter + 1, method, L, cputime))
            else:
                self.print("Iteration %d: loglike=%e (%.3f seconds)"
                           % (self.iter + 1, L, cputime))
        self.converged = False
        if not self.ignore_bound_checks and not self.annealing_changed and self.iter > 0:
            L0 = self.L[self.iter - 1]
            if L0 - L > 1e-6:
                warnings.warn("Lower bound decreased %e! Bug somewhere or "
                              "numerical inaccuracy?" % (L0 - L))
            if tol is None:
                tol = self.tol
            div = 0.5 * (abs(L0) + abs(L))
            if (L - L0) / div < tol:
                if verbose:
                    self.print("Converged at iteration %d." % (self.iter + 1))
                self.converged = True
        self.iter += 1
        self.annealing_changed = False
        # auto-save (vmp.py:749-758): here, so that every iteration driver (update, optimize,
        # pattern_search) and the converging iteration are covered
        if self.autosave_iterations > 0 and self.iter % self.autosave_iterations == 0:
            if self.autosave_nodes is not None:
                self.save(*self.autosave_nodes, filename=self._autosave_target())
            else:
                self.save(filename=self._autosave_target())
            if verbose:
                self.print('Auto-saved to %s' % self.autosave_filename)
        return self.converged
