"""CPU oracle for the FDEM forward + likelihood path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``geobipy_amd``) never does; it fails loudly without its HIP library.
"""
