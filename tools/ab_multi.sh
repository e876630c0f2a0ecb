#!/bin/bash
# Same-box A/B of OE.multiple_reflect (toroid, whispering-gallery bounces; tools/probe_multi.py):
# alternates the libraries given as arguments ("" = the built one), three rounds.
#   gpurun -- 'bash tools/ab_multi.sh "" xrt_amd/ab/libxrt_old.so'
cd "$GRAFT_REPO_ROOT"
N=${AB_MULTI_RAYS:-1e7}
for ROUND in 1 2 3; do
  for LIB in "$@"; do
    XRT_HIP_LIBRARY=$LIB PYTHONPATH=.:tests timeout 300 python tools/probe_multi.py $N 3 2>&1 | grep -v amdgpu.ids | grep -i "ms\b\|bounce" | sed "s|^|[$LIB] |" | cut -c1-260
  done
done
