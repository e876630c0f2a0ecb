#!/bin/bash
# What the plot's part of the ray kernel costs (reflect_fused_gen_scr_plot), piece by piece: variants of
# the library without the wave's sort + stores, without the sort alone, without weight and bins
#   for v in NO_EMIT NO_SORT NO_TAKE; do tools/build_variant.sh tail_$v "-DTAIL_AB_$v" reflect_hot_plot; done
#   gpurun -- 'bash tools/ab_tail_pass.sh'
# (the variants' plots are wrong: timing only). Same box, rocprofv3 averages. CAUTION: NO_EMIT and NO_SORT leave
# plot_tail_tiles with run tables that do not describe the records -- it walked them for the whole 900 s of a
# gpurun call in round 6; NO_TAKE (every ray in the bucket of the unselected) is consistent and safe.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "" xrt_amd/ab/libxrt_tail_NO_EMIT.so xrt_amd/ab/libxrt_tail_NO_SORT.so xrt_amd/ab/libxrt_tail_NO_TAKE.so; do
  [ -z "$L" ] || [ -f "$L" ] || continue
  echo "=== ${L:-default}"
  rm -rf /tmp/hp
  XRT_HIP_LIBRARY=$L timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/hp -o hp -- env PYTHONPATH=. python tools/probe_plot_tail.py 1e7 5 > /tmp/hp.log 2>&1
  python tools/prof_stats.py /tmp/hp 14 2>&1 | grep "gen_scr"
done
