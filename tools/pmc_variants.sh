#!/bin/bash
# VALU instructions per wave of the reflect kernels for several builds of the library:
#   gpurun -- 'bash tools/pmc_variants.sh "" xrt_amd/ab/libxrt_X.so ...'
for LIB in "$@"; do
  echo "== ${LIB:-built library}"
  XRT_HIP_LIBRARY=$LIB bash tools/pmc_valu.sh | grep reflect
done
