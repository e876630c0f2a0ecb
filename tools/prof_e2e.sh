#!/bin/bash
# per-kernel times of the end-to-end run_ray_tracing leg (bench.py e2e) under rocprofv3
#   bash tools/prof_e2e.sh [rays]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/e2e
cat > /tmp/e2e_run.py <<'PY'
import sys, json, bench
r = bench.bench_e2e(int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000, repeats=20)
r.pop('host_source', None)
print(json.dumps(r))
PY
rocprofv3 --kernel-trace --stats -d /tmp/e2e -o e2e -- env PYTHONPATH=. XRT_E2E_NO_HOST=1 XRT_E2E_NO_SMALL=1 python /tmp/e2e_run.py ${1:-1e7} 2> /tmp/e2e.err | tail -1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/e2e/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels group by name, grid_x order by sum(duration) desc").fetchall()
tot = sum(r[4] for r in rows)
for r in rows[:16]:
    print("%-64s grid %9d  n %4d  avg %8.1f us  %5.1f %%" % (r[0].split('(')[0][:64], r[1], r[2], r[3] / 1e3, 100. * r[4] / tot))
PY
