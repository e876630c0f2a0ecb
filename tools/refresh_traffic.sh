#!/bin/bash
# Runs on the GPU box (via gpurun): the four HBM counter passes of tools/profile_workload.py alone
# (FETCH_SIZE / WRITE_SIZE, the full passes and the --nolocal ones) -> profiles/hbm_traffic.json
#   gpurun --timeout 900 -- 'bash tools/refresh_traffic.sh'
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/traffic
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch -- python tools/profile_workload.py 1e7 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write -- python tools/profile_workload.py 1e7 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_nl -o fetch -- python tools/profile_workload.py 1e7 --nolocal > $O/fetch_nl.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_nl -o write -- python tools/profile_workload.py 1e7 --nolocal > $O/write_nl.log 2>&1
F=$(find $O/fetch -name '*.db' | head -1); W=$(find $O/write -name '*.db' | head -1)
FN=$(find $O/fetch_nl -name '*.db' | head -1); WN=$(find $O/write_nl -name '*.db' | head -1)
python tools/rocpd_summary.py tmp - "$F" "$W" 1e7 "$FN" "$WN"
cp profiles/hbm_traffic.json $O/
find $O -name '*.db' -size +40M -delete
