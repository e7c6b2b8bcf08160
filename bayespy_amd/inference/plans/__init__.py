"""
Compiled execution plans: each plan owns the HBM-resident state of one model
block and maps the reference's per-node operations (``update``,
``lower_bound_contribution``, ``get_moments``) onto HIP kernel launches.
"""
from .pca import PCAPlan
from .masked_pca import MaskedPCAPlan
from .gmm import GMMPlan
from .lssm import LSSMPlan
from .lssm_masked import MaskedLSSMPlan

PLAN_TYPES = [PCAPlan, MaskedPCAPlan, GMMPlan, LSSMPlan, MaskedLSSMPlan]


def _reusable_plans(nodes, engine, options=None):
    """In the reference the state of q lives in the nodes, so a second ``VB`` over nodes that
    were already updated continues from their posteriors.  Here it lives in the plans: when the
    plans that own these nodes cover exactly what is asked for, they are kept."""
    from ...nodes.node import Stochastic
    from .generic import GenericPlan
    plans = []
    for n in nodes:
        if not isinstance(n, Stochastic):
            continue
        p = n._plan
        if p is None:
            return None
        if not any(p is q for q in plans):
            plans.append(p)
    if not plans:
        return None
    if engine == 'generic' and not all(isinstance(p, GenericPlan) for p in plans):
        return None
    covered = set(id(m) for p in plans for m in p.nodes())
    if not all(id(n) in covered for n in nodes):
        return None
    if engine != 'generic' and all(isinstance(p, GenericPlan) for p in plans) \
            and getattr(plans[0], '_engine_request', None) == 'generic':
        # built on request for engine='generic'; without that request the fused blocks get their
        # chance again (the generic engine was kept silently)
        import warnings
        warnings.warn("the existing plan was built with engine='generic'; this VB matches the "
                      "fused blocks again (posterior state starts again from the nodes' "
                      "initialisation)")
        return None
    for p in plans:
        for key, val in (options or {}).items():
            have = {'stats': getattr(p, 'stats', None), 'layout': getattr(p, 'plate_layout', None),
                    'chunk': getattr(p, 'chunk', None)}.get(key, val)
            if val is not None and have is not None and have != val:
                import warnings
                warnings.warn('the existing plan was built with %s=%r; %s=%r asks for a new one '
                              '(posterior state starts again from the nodes\' initialisation)'
                              % (key, have, key, val))
                return None
    return plans


def compile_model(nodes, engine=None, **options):
    """Cover the stochastic nodes of ``nodes`` with plans.  Raises
    NotImplementedError (loudly -- there is no CPU fallback) when a node is not
    covered by any built plan."""
    from ...nodes.node import Stochastic
    import os
    if engine is None:
        engine = os.environ.get('BAYESPY_AMD_ENGINE', 'auto')
    kept = _reusable_plans(nodes, engine, options)
    if kept is not None:
        return kept
    stale = [n.name for n in nodes if isinstance(n, Stochastic) and n._plan is not None
             and getattr(n._plan, 'has_state', lambda: False)()]
    if stale:
        import warnings
        warnings.warn('nodes %s already hold posterior state in another execution plan; the new '
                      'plan starts from their initialisation' % ', '.join(stale))
    if engine == 'generic':
        from .generic import GenericPlan
        plan = GenericPlan(nodes)
        plan._engine_request = 'generic'
        return [plan]
    remaining = [n for n in nodes]
    plans = []
    progress = True
    while progress:
        progress = False
        for P in PLAN_TYPES:
            roles = P.match(remaining)
            if roles is not None:
                plan = P(roles, **options)
                plans.append(plan)
                used = set(id(n) for n in roles.values())
                remaining = [n for n in remaining if id(n) not in used]
                progress = True
                break
    left = [n for n in remaining if isinstance(n, Stochastic)]
    if left:
        # nodes outside the fused blocks: the whole model runs on the generic device
        # message-passing engine (raises NotImplementedError for unknown node types).  A model
        # that RESEMBLES a fused block but misses its matcher is told why: the generic engine
        # keeps the reference's per-plate arrays ((N, K, K) second moments, (N, K, D, D) mixture
        # intermediates), which is the difference between milliseconds and out-of-memory at
        # large N.
        why = []
        for P in PLAN_TYPES:
            try:
                P.match(remaining, why)
            except TypeError:       # a test double without the diagnostic argument
                pass
        if why:
            import warnings
            warnings.warn('this model runs on the generic message-passing engine, not on a fused '
                          'block -- ' + '; '.join(dict.fromkeys(why))
                          + ' (engine="generic" selects this engine silently)',
                          stacklevel=3)
        from .generic import GenericPlan
        return [GenericPlan(nodes)]
    return plans
