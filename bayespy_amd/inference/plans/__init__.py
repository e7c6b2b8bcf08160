"""
Compiled execution plans: each plan owns the HBM-resident state of one model
block and maps the reference's per-node operations (``update``,
``lower_bound_contribution``, ``get_moments``) onto HIP kernel launches.
"""
from .pca import PCAPlan
from .masked_pca import MaskedPCAPlan
from .gmm import GMMPlan
from .lssm import LSSMPlan

PLAN_TYPES = [PCAPlan, MaskedPCAPlan, GMMPlan, LSSMPlan]


def compile_model(nodes, engine=None, **options):
    """Cover the stochastic nodes of ``nodes`` with plans.  Raises
    NotImplementedError (loudly -- there is no CPU fallback) when a node is not
    covered by any built plan."""
    from ...nodes.node import Stochastic
    import os
    if engine is None:
        engine = os.environ.get('BAYESPY_AMD_ENGINE', 'auto')
    if engine == 'generic':
        from .generic import GenericPlan
        return [GenericPlan(nodes)]
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
        # message-passing engine (raises NotImplementedError for unknown node types)
        from .generic import GenericPlan
        return [GenericPlan(nodes)]
    return plans
