"""ORACLE — test infrastructure only.

CPU restatements of the reference's hot path used as the *checker*:
``warp_splat_ref.c`` (C, bit-exact fp32) and ``hardnet_ref.py`` (torch fp32).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The product (``panoptic-forecasting_amd/``, libpfhip.so)
never does and fails loudly when its HIP library is missing.
"""
