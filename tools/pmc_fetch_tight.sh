#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of reflect_fused with every ray hitting the mirror (--tight) against
# the cfg2 beam (2.3 % lost / over rays: 77 % of the waves hold one and store in two parts).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
for OPT in "" "--tight"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcv
    rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcv -o p -- python tools/profile_workload.py 1e7 --no-kirchhoff $OPT > /dev/null 2>&1
    python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmcv/**/*.db', recursive=True)
c = sqlite3.connect(db[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t.startswith('counters_collection')][0]
for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from %s where kernel_name like '%%reflect_fused<%%' group by kernel_name, counter_name" % view):
    print('[$OPT]', r[0][:50], r[1], '%.6g KB' % r[2], r[3])
PY
  done
done
