#!/bin/bash
# Same-box A/B of the passes outside the headline kernels -- layered materials, figure errors, user
# surfaces' generic family, single crystals (tools/probe_multilayer.py, probe_figure.py,
# probe_exact_pass.py): alternates the libraries given as arguments ("" = the built one), two rounds.
#   gpurun -- 'bash tools/ab_families.sh "" xrt_amd/ab/libxrt_early.so'
cd "$GRAFT_REPO_ROOT"
for ROUND in 1 2; do
  for LIB in "$@"; do
    echo "=== [$LIB]"
    XRT_HIP_LIBRARY=$LIB PYTHONPATH=.:tests timeout 300 python tools/probe_multilayer.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
    XRT_HIP_LIBRARY=$LIB PYTHONPATH=.:tests timeout 300 python tools/probe_figure.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done
