#!/bin/bash
# ISA of the hot reflect kernels with line tables, for tools/isa_lines.py:
#   tools/isa_hot.sh [unit, default reflect_hot]  ->  /tmp/isa/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s
set -e
U=${1:-reflect_hot}
mkdir -p /tmp/isa
cd "$(dirname "$0")/../xrt_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden \
      -mllvm -instcombine-max-copied-from-constant-users=100000 \
  -gline-tables-only -c $U.hip -o /tmp/isa/$U.o --save-temps=obj 2>/dev/null
echo /tmp/isa/$U-hip-amdgcn-amd-amdhsa-gfx950.s
