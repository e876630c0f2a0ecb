#!/bin/bash
# plot_tail_tiles with its LDS additions switched off in turn (tools/build_variant.sh
# tail_noplanes "-DTAIL_AB_NO_PLANES" hist; tail_none "-DTAIL_AB_NO_CLINES -DTAIL_AB_NO_PLANES"):
# what of the kernel's time the additions are. Same box, rocprofv3 averages over
# tools/probe_plot_tail.py (focused and wide plots mixed).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "" xrt_amd/ab/libxrt_tail_noplanes.so xrt_amd/ab/libxrt_tail_none.so; do
  [ -z "$L" ] || [ -f "$L" ] || continue
  echo "=== ${L:-default}"
  rm -rf /tmp/hp
  XRT_HIP_LIBRARY=$L rocprofv3 --kernel-trace --stats -d /tmp/hp -o hp -- env PYTHONPATH=. python tools/probe_plot_tail.py 1e7 5 > /tmp/hp.log 2>&1
  python tools/prof_stats.py /tmp/hp 12 2>&1 | grep "plot_tail\|gen_scr_plot\|reduce"
done
