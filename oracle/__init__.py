"""
CPU oracle for the Graph-KIR typing hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain Python / NumPy, the algorithm of the reference
(`linnil1/KIR_graph`, ``graphkir/*.py``) for the path this repository
accelerates.  Every function cites the reference lines it follows.

Rules (task section 3):
* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it -- as the checker, never as the product;
* ``kir_graph_amd`` never imports ``oracle``; the product path fails loudly if
  the HIP library is missing instead of falling back to this code.

Pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned by fixtures in ``tests/golden/``
that were produced by importing the reference itself in the build container
(``tests/golden/make_golden.py``; numpy 2.2.6), and
``tests/test_oracle_golden.py`` checks every oracle function against them.

Floating point caveat: ``numpy.log10`` and the tie order of ``numpy.argsort``
depend on the host's SIMD level (SVML on AVX-512 hosts, libm elsewhere), so
the reference's own last bits and tie breaks are host dependent.  The oracle
calls the same NumPy primitives as the reference, hence it reproduces the
reference *on the host it runs on*; fixtures are compared with a 1e-9
relative tolerance on floats and exactly on integers, ids and allele calls.
"""
