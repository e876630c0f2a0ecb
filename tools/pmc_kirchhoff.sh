#!/bin/bash
# PMC passes over the Kirchhoff stream kernel; prints per-kernel averages per launch.
#   bash tools/pmc_kirchhoff.sh cfg4      the cfg4 shape (fast loop, 1e6 x 512^2)
#   bash tools/pmc_kirchhoff.sh general   the general loop gen_sp_n (2e5 x 2e5, bench leg kirchhoff_general)
# Counters only with --kernel-trace (no other trace domain), one small group per pass.
WHICH=${1:-cfg4}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
cat > /tmp/kg_run.py <<'PY'
import numpy as np, torch
from xrt_amd import hipcalls, workloads
h = workloads.kirchhoff_general()
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = [up(h[f]) for f in ('px', 'py', 'pz', 'sx', 'sy', 'sz', 'nx', 'ny', 'nz', 'nl', 'k', 'Es', 'Ep')]
for _ in range(2):
    hipcalls.kirchhoff(*args)
torch.cuda.synchronize()
print(sorted(hipcalls.kirchhoff_report()['variants']), h['ns'], h['npix'])
PY
if [ "$WHICH" = general ]; then CMD="python /tmp/kg_run.py"; PAIRS=4e10; else CMD="python tools/probe_kirchhoff.py 4 0"; PAIRS=2.62144e11; fi
echo "# $WHICH: $CMD  (pairs per launch $PAIRS)"
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc -o p -- $CMD > /tmp/pmc_run.log 2>&1 || tail -3 /tmp/pmc_run.log
  PAIRS=$PAIRS python - <<PY
import sqlite3, glob, os
db = glob.glob('/tmp/pmc/**/*.db', recursive=True)
if not db:
    print('no db for $C')
else:
    c = sqlite3.connect(db[0])
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%kirchhoff_stream%' group by kernel_name, counter_name").fetchall()
        for r in rows: print('%-24s %.4g  (n=%d)  per pair %.4g' % (r[1], r[2], r[3], r[2] / float(os.environ['PAIRS'])))
    except Exception as e:
        print('ERR', e)
PY
done
