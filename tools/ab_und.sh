#!/bin/bash
# same-box A/B of the undulator map kernel: the regular library and the variants given
#   gpurun -- 'bash tools/ab_und.sh xrt_amd/ab/libxrt_A.so xrt_amd/ab/libxrt_B.so'
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
cat > /tmp/und_run.py <<'PY'
import bench, torch, json
for k in range(3):
    r, _ = bench.bench_undulator(with_cpu=False) if isinstance(bench.bench_undulator(with_cpu=False), tuple) else (bench.bench_undulator(with_cpu=False), None)
    print('   ms %.4f frac %.3f' % (r['ms'], r['roofline']['frac']))
PY
for lib in "" "$@"; do
  echo "== ${lib:-regular}"
  XRT_HIP_LIBRARY=$lib python /tmp/und_run.py 2>&1 | grep -v amdgpu.ids | tail -3
done
