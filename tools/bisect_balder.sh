#!/bin/bash
# Which leg of bench.py disturbs the Balder leg? (same box, full runs)
cd "$GRAFT_REPO_ROOT"
run() { L=$1; shift; XRT_HIP_LIBRARY=$L python bench.py "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('[$L] [$*] balder %.3f ms' % (d['balder']['seconds'] * 1e3))"; }
run ""
run xrt_amd/ab/libxrt_limit300.so
run "" --skip-undulator
run "" --skip-kirchhoff
