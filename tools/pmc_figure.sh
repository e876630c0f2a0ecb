#!/bin/bash
# SQ counters of the Figured<0> fused kernel (cfg2 toroid + a 512 x 128 roughness map, 1e7 rays).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.:tests
cat > /tmp/figure_job.py <<'PY'
import numpy as np, torch
import figure_cases as fc
from xrt_amd import workloads
n = 10_000_000
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
g = np.load(fc.GOLDEN + '/g2_figure_toroid.npz')
oe = fc.element('g2_figure_toroid', g)
out = None
for _ in range(4):
    out = oe.reflect(beam, out=out)
torch.cuda.synchronize()
PY
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc -o p -- python /tmp/figure_job.py > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc/**/*.db', recursive=True)
if not db:
    print('no db for $C')
else:
    c = sqlite3.connect(db[0])
    try:
        rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%reflect_fused%' group by kernel_name, counter_name").fetchall()
        for r in rows: print(r[0][:60], r[1], '%.4g' % r[2], r[3])
    except Exception as e:
        print('ERR', e)
PY
done
