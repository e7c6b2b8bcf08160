"""
oracle/ -- CPU restatement of the reference's VMP hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``bayespy_amd``) may import this package.  The
only legal importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker, never as
the thing that is measured or shipped.

Parity status: PINNED.  Every function here is checked against the live
reference (bayespy imported from /root/reference in the authoring container,
see ``oracle/make_golden.py``) through the golden fixtures committed under
``tests/golden/`` and through the reference's own known-answer vectors
(quickstart ELBO trace, doc/source/user_guide/quickstart.rst:111-118).
"""
