#!/bin/bash
# per-launch times of the histogram kernels under rocprofv3 (tools/probe_hist.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/hp
rocprofv3 --kernel-trace --stats -d /tmp/hp -o hp -- env PYTHONPATH=. python tools/probe_hist.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/hp/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
for r in c.execute("select name, grid_x, lds_size, count(*), avg(duration) from kernels where name like '%hist%' or name like '%fill%' or name like '%memset%' group by name, grid_x, lds_size order by name, lds_size"):
    print(r[0][:44], r[1], r[2], r[3], "%.1f us" % (r[4] / 1e3))
PY
