"""Balder chain alone (bench.bench_balder at 1e7 rays), for rocprofv3 runs."""
import json, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
r = bench.bench_balder(10_000_000, runs=5)
bench.bench_balder(10_000_000, runs=1, both=False)     # (the last kernels of a trace: one pass, on demand)
print(json.dumps(r))
