#!/bin/bash
# Same-box A/B of the cfg2 / cfg3 passes: alternates the libraries given as arguments
# ("" = the built one), three rounds; prints kernel_ms / pass_ms / DCM ms per run.
#   gpurun -- 'bash tools/ab_reflect.sh "" xrt_amd/ab/libxrt_old.so'
cd "$GRAFT_REPO_ROOT"
for ROUND in 1 2 3; do
  for LIB in "$@"; do
    XRT_HIP_LIBRARY=$LIB python bench.py --steps 40 --warmup 5 --skip-kirchhoff --skip-undulator \
      --skip-softimax --skip-cpu-baseline --skip-balder --skip-e2e $AB_EXTRA 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('[%s] kernel %.4f ms  pass %.4f ms  step %.4f ms  dcm %.4f ms' % ('$LIB', d['kernel_ms'], d['pass_ms'], d['ms_per_step'], d.get('dcm', {}).get('ms_per_step', float('nan'))))"
  done
done
