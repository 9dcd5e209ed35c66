"""CPU oracle for the vr180-convert hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package ``vr180_convert_amd`` never does.
"""
