"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (numpy/scipy) restatement of GetDist 1.7.7's weighted-statistics + 1D/2D KDE hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from here, and only as the checker / reported CPU baseline -- never as the product path.
The shipped package (``getdist_amd``) must not import this package.

Parity status: PINNED.  ``oracle/validate_against_reference.py`` (run in the build container, where
``/root/reference`` exists) compares every function here against the imported reference, and
``tests/golden/make_golden.py`` stores reference outputs as fixtures under ``tests/golden/`` which
``tests/test_oracle_golden.py`` re-checks on any box.
"""
