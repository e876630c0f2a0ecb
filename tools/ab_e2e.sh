#!/bin/bash
# Same-box A/B of the end-to-end run_ray_tracing iteration (device source -> toroid -> screen ->
# 256 x 256 plot, 1e7 rays; the plot in the tail of the pass and as launches of its own):
# alternates the libraries given as arguments ("" = the built one), three rounds.
#   gpurun -- 'bash tools/ab_e2e.sh "" xrt_amd/ab/libxrt_old.so'
cd "$GRAFT_REPO_ROOT"
for ROUND in 1 2 3; do
  for LIB in "$@"; do
    echo "[$LIB] $(XRT_HIP_LIBRARY=$LIB PYTHONPATH=. python tools/probe_plot_tail.py 1e7 20 2>/dev/null | grep -i 'focused' | tr '\n' ' ')"
  done
done
