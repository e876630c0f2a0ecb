#!/bin/bash
# One PMC pass: VALU instructions per wave of the reflect kernels (cfg2 / cfg3 shapes).
#   gpurun -- 'bash tools/pmc_valu.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
rm -rf /tmp/pmc
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_TRANS -d /tmp/pmc -o p -- python tools/profile_workload.py 1e7 --no-kirchhoff > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc/**/*.db', recursive=True)
c = sqlite3.connect(db[0])
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%reflect_fused%' or kernel_name like '%und_imap%' group by kernel_name, counter_name").fetchall()
d = {}
for r in rows:
    d.setdefault(r[0], {})[r[1]] = r[2]
for k, v in d.items():
    w = v.get('SQ_WAVES', 0) or 1
    print('%-70s VALU/wave %.0f  trans/wave %.0f  waves %.0f' % (k[:70], v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_VALU_TRANS', 0) / w, w))
PY
