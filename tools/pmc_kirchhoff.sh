#!/bin/bash
# PMC passes over the Kirchhoff stream kernel (cfg4 shape); prints per-kernel averages.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc -o p -- python tools/probe_kirchhoff.py 4 0 > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc/**/*.db', recursive=True)
if not db:
    print('no db for $C')
else:
    c = sqlite3.connect(db[0])
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%kirchhoff_stream%' group by kernel_name, counter_name").fetchall()
        for r in rows: print(r[1], '%.4g' % r[2], r[3])
    except Exception as e:
        print('ERR', e)
PY
done
