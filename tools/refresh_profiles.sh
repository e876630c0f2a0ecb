#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the default bench.py
# and the two HBM counter passes on tools/profile_workload.py; summaries go to
# profiles/ (copied back through gpurun_out/).
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh 01'
set -u
RND=${1:-01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r$RND
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python bench.py > $O/bench_under_rocprof.json 2> $O/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch -- python tools/profile_workload.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write -- python tools/profile_workload.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_nl -o fetch -- python tools/profile_workload.py 1e7 --nolocal > $O/fetch_nl.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_nl -o write -- python tools/profile_workload.py 1e7 --nolocal > $O/write_nl.log 2>&1
FN=$(find $O/fetch_nl -name '*.db' | head -1); WN=$(find $O/write_nl -name '*.db' | head -1)
T=$(find $O/trace -name '*.db' | head -1); F=$(find $O/fetch -name '*.db' | head -1); W=$(find $O/write -name '*.db' | head -1)
echo "dbs: $T $F $W"
python tools/rocpd_summary.py $RND "$T" "$F" "$W" 1e7 "$FN" "$WN"
mkdir -p $O/summaries && cp profiles/r${RND}_kernel_stats.csv profiles/hbm_traffic.json $O/summaries/
tail -c 600 $O/bench_under_rocprof.json | head -c 600; echo
# SQ counters of the reflect kernels and the stand-alone probes the DESIGN quotes
bash tools/pmc_reflect.sh > profiles/r${RND}_reflect_pmc.txt 2>&1
bash tools/pmc_und.sh > profiles/r${RND}_und_pmc.txt 2>&1
bash tools/prof_hist.sh > profiles/r${RND}_hist_kernels.txt 2>&1
# round 4: SQ counters of BOTH Kirchhoff loops (cfg4 fast loop, general gen_sp_n), HBM bytes of
# the histogram kernels (also profiles/hist_traffic.json, read by bench.py), the kernels of one
# end-to-end run_ray_tracing iteration
bash tools/pmc_kirchhoff.sh cfg4 > profiles/r${RND}_kirchhoff_pmc.txt 2>&1
bash tools/pmc_kirchhoff.sh general > profiles/r${RND}_kirchhoff_general_pmc.txt 2>&1
bash tools/pmc_hist.sh > profiles/r${RND}_hist_pmc.txt 2>&1
bash tools/prof_e2e.sh 1e7 > profiles/r${RND}_e2e_kernels.txt 2>&1
# small beams: the kernels of the same iteration at 2e3 / 1e5 / 1e6 rays (the fixed cost of every
# launch), and the host time of every element call
for n in 2e3 1e5 1e6; do echo "== $n rays"; bash tools/prof_e2e.sh $n 2>&1 | grep -v '^{'; done \
  > profiles/r${RND}_e2e_small_kernels.txt
PYTHONPATH=. python tools/probe_host_overhead.py 2>&1 | grep 'us per call' > profiles/r${RND}_host_overhead.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/probe_fp64_rates.hip -o /tmp/probe_fp64_rates 2>/dev/null && \
  timeout 300 /tmp/probe_fp64_rates > profiles/r${RND}_probe_fp64_rates.txt 2>&1
for P in probe_stream probe_occupancy probe_atomics probe_fp64_seeds probe_lds_atomics; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/$P.hip -o /tmp/$P 2>/dev/null && \
    timeout 300 /tmp/$P > profiles/r${RND}_$P.txt 2>&1
done
hipcc --offload-arch=gfx950 -O3 tools/probes/probe_stream_wide.hip -o /tmp/probe_stream_wide 2>/dev/null && \
  timeout 300 /tmp/probe_stream_wide > profiles/r${RND}_probe_stream_wide.txt 2>&1
# (OE(figureError=...) is frozen since round 4: profiles/r04_figure_pass.txt, r04_figure_pmc.txt)
# round 5: OE.multiple_reflect (all bounces of one call), the e2e iteration at three beam sizes
# eager / graph with the fusion and the small-beam histogram switched off in turn
PYTHONPATH=.:tests timeout 300 python tools/probe_multi.py 1e6 5 2>&1 | grep -v amdgpu.ids > profiles/r${RND}_multiple_reflect.txt
PYTHONPATH=.:tests timeout 300 python tools/probe_multi.py 1e7 3 2>&1 | grep -v amdgpu.ids >> profiles/r${RND}_multiple_reflect.txt
for env in "" "XRT_HIP_HIST_NO_SMALL=1" "XRT_HIP_NO_FUSE=1"; do echo "== [$env]"; env PYTHONPATH=. $env python tools/probe_e2e_sizes.py 200 2>&1 | grep rays; done > profiles/r${RND}_e2e_sizes.txt
# the kernels of the Balder chain (bench.py balder leg) in launch order, one pass of the beam
( cd /tmp && rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats -d /tmp/pb -o pb -- python $GRAFT_REPO_ROOT/tools/probe_balder.py > /tmp/pb.log 2>&1 )
{ grep '^{' /tmp/pb.log | cut -c1-400; python tools/prof_sequence.py /tmp/pb 30; } > profiles/r${RND}_balder_kernels.txt 2>&1
# round 6: the Balder chain's fused pieces against their separate launches -- the focusing mirror
# with its two slits and the sample screen in the tail of its pass; both faces of the filter in one kernel
{ python tools/probe_front_end.py; python tools/probe_tail_apertures.py; python tools/probe_xtal_tail.py; python tools/probe_plate2.py; \
  XRT_HIP_DCM_TWO_PASSES=1 python tools/probe_plate2.py | sed 's/\[\]/[two passes]/'; } 2>&1 \
  | grep -v amdgpu.ids > profiles/r${RND}_balder_tails.txt
# round 6: the plot in the tail of the pass -- HBM bytes of an e2e iteration either way
# (profiles/plot_tail_traffic.json, read by bench.py), ms per iteration with focused / wide plot limits
bash tools/pmc_plot_tail.sh > profiles/r${RND}_plot_tail_pmc.txt 2>&1
{ python tools/probe_plot_tail.py 1e7 20; python tools/probe_plot_tail.py 1e5 200; } 2>&1 | grep -v amdgpu.ids > profiles/r${RND}_plot_tail.txt
cp profiles/plot_tail_traffic.json $O/summaries/ 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -DNOUT=40 -DBLOCK=256 tools/probes/probe_occupancy.hip -o /tmp/po40 2>/dev/null && \
  timeout 300 /tmp/po40 > profiles/r${RND}_probe_occupancy_dcm_shape.txt 2>&1
cp profiles/r${RND}_*.txt profiles/hist_traffic.json $O/summaries/ 2>/dev/null
find $O -name '*.db' -size +40M -delete
