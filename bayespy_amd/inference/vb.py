"""
The VB engine: iteration driver, lower-bound bookkeeping, convergence test.

Host-side mirror of ``bayespy.inference.vmp.vmp.VB`` (vmp.py:52-172,
:180-199, :693-764): same constructor/``update`` signature, same iteration
semantics (update the given nodes in order, then evaluate the full lower
bound, warn if it decreased by more than 1e-6, stop when the relative change
drops below ``tol``), same log line.  The numbers come from the compiled
plans, i.e. from HIP kernels.
"""
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
        self.autosave_filename = autosave_filename
        self.autosave_iterations = int(autosave_iterations or 0)
        if self.autosave_iterations and not autosave_filename:
            raise ValueError('autosave_iterations needs autosave_filename')
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

    def set_callback(self, callback):
        self.callback = callback

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
                if self.autosave_iterations and self.iter % self.autosave_iterations == 0:
                    self.save()
        finally:
            # plans may keep plate-sized work in flight on their own streams across
            # iterations; order the caller's stream after it before handing back control
            for p in self.plans:
                fin = getattr(p, 'finish', None)
                if fin is not None:
                    fin()

    # -- persistence (vmp.py:237-356) ------------------------------------------------------
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
        w = Writer(filename)
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
        w.close()

    def load(self, *nodes, filename=None, nodes_only=False):
        from .checkpoint import Reader
        nodes = [self[n] for n in nodes if n is not None] if nodes else list(self.model)
        filename = filename or self.autosave_filename
        if not filename:
            raise Exception("Filename must be given.")
        r = Reader(filename)
        try:
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

    # -- lower bound (vmp.py:180-199) ----------------------------------------------------
    def compute_lowerbound(self, ignore_masked=True):
        return sum(n.lower_bound_contribution() for n in self.model)

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
        if not self.ignore_bound_checks and self.iter > 0:
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
        return self.converged
