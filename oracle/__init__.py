"""TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy / plain C) of the xrt reference algorithms for the hot
path. Nothing in ``xrt_amd/`` may import this package: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline, never as the shipped compute path.
"""
