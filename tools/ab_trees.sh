#!/bin/bash
# Same-box A/B of two source trees (each with its own library and host code): the repo and
# a copy of an earlier commit under ab_oldtree/. Prints kernel / pass / DCM times.
cd "$GRAFT_REPO_ROOT"
for ROUND in 1 2 3; do
  for T in . ab_oldtree; do
    (cd $T && python bench.py --steps 40 --warmup 5 --skip-kirchhoff --skip-undulator \
      --skip-softimax --skip-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('[$T] kernel %.4f ms  pass %.4f ms  overhead %.1f us  step %.4f ms  dcm %.4f ms' % (d['kernel_ms'], d['pass_ms'], (d['pass_ms'] - d['kernel_ms']) * 1e3, d['ms_per_step'], d.get('dcm', {}).get('ms_per_step', float('nan'))))")
  done
done
