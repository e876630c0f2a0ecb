#!/bin/bash
# per-launch times of the small kernels around the fused ray kernels (decide, verdict):
#   gpurun -- 'bash tools/prof_small.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/dq
rocprofv3 --kernel-trace --stats -d /tmp/dq -o dq -- python bench.py --skip-kirchhoff --skip-undulator --skip-softimax --skip-balder --skip-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/dq/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for r in c.execute("select name, count(*), avg(duration) from kernels where name like '%decide%' or name like '%reflect_exact%' or name like '%dcm_exact%' or name like '%reflect_fused%' group by name"):
    print(r[0][:70], r[1], '%.1f us' % (r[2] / 1e3))
PY
