set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06b
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python bench.py > $O/bench_under_rocprof.json 2> $O/trace.log
python bench.py > $O/r06_bench.json 2> $O/bench.err
( cd /tmp && rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats -d /tmp/pb -o pb -- python $GRAFT_REPO_ROOT/tools/probe_balder.py > /tmp/pb.log 2>&1 )
{ grep '^{' /tmp/pb.log | cut -c1-400; python tools/prof_sequence.py /tmp/pb 30; } > $O/r06_balder_kernels.txt 2>&1
find $O -name '*.db' -size +60M -delete
tail -c 300 $O/bench_under_rocprof.json
