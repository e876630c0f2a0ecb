#!/bin/bash
# Same-box A/B of the Balder leg (six surfaces, 1e7 rays): alternates the libraries given as
# arguments ("" = the built one), three rounds.
#   gpurun -- 'bash tools/ab_balder.sh "" xrt_amd/ab/libxrt_old.so'
cd "$GRAFT_REPO_ROOT"
for ROUND in 1 2 3; do
  for LIB in "$@"; do
    XRT_HIP_LIBRARY=$LIB python bench.py --steps 5 --warmup 2 --skip-kirchhoff --skip-undulator \
      --skip-softimax --skip-cpu-baseline --skip-e2e --skip-dcm 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('[%s] balder %.3f ms' % ('$LIB', d['balder']['seconds'] * 1e3))"
  done
done
