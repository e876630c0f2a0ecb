#!/bin/bash
# A variant of the library for same-box A/B runs: recompiles the given units (default: the hot
# reflect kernels) with extra flags and links them with the objects of the regular build.
#   tools/build_variant.sh NAME "-DXRT_SOMETHING=1" [unit ...]   ->  xrt_amd/ab/libxrt_NAME.so
#   gpurun -- 'bash tools/ab_reflect.sh "" xrt_amd/ab/libxrt_NAME.so'
set -e
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-reflect_hot}
# (ELIDE= in the environment: without the raised elision limit of csrc/build.py)
ELIDE=${ELIDE--mllvm -instcombine-max-copied-from-constant-users=100000}
cd "$(dirname "$0")/../xrt_amd/csrc"
V=/tmp/xrt_var_$NAME   # variant objects stay out of the tree
mkdir -p $V ../ab
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  if [[ " $UNITS " == *" $b "* ]]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden \
      $ELIDE \
      -Wno-unused-function $FLAGS -c $b.hip -o $V/$b.o &
    OBJS="$OBJS $V/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libxrt_$NAME.so $OBJS
echo xrt_amd/ab/libxrt_$NAME.so
