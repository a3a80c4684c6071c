"""OUTSIDE the scope of this repository's contract (§2 rows 3, 4, 12, 13 are marked out of scope: the solvers
and the result / attribute glue of ``/root/reference/src/infiniteopt_backend.jl:159-508`` and ``ext/*.jl``).

Kept, frozen, for one reason: ``tests/test_gpu_solve.py`` and ``tests/test_known_answers.py`` drive the device evaluator
through these small solvers to the constants the reference's own tests assert (``test/solve.jl:146,154,187,206``,
``test/ipopt.jl:180-181``) — the only reference-held pins that reach the Jacobian and the Hessian.  Nothing here is on the
hot path, nothing in the package imports it eagerly, and nothing more is built on it:

* ``newton.py``  Lagrange-Newton for equality-constrained models (over the chain KKT solver)
* ``ipm.py``     a compact restatement of Ipopt's published algorithm (no restoration phase — by decision)
* ``results.py`` status tables / result objects that ``backend.py``'s result queries read
"""
