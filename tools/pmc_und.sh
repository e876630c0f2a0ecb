#!/bin/bash
# SQ counters of the undulator map kernel (the bench's und_imap launch), two passes:
#   gpurun -- 'bash tools/pmc_und.sh'
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
cat > /tmp/und_run.py <<'PY'
import bench, torch
bench.bench_undulator(with_cpu=False)
torch.cuda.synchronize()
PY
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  rm -rf /tmp/pmc
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc -o p -- python /tmp/und_run.py > /tmp/und_run.log 2>&1 || tail -5 /tmp/und_run.log
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmc/**/*.db', recursive=True)
c = sqlite3.connect(db[0])
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%und_imap%' group by kernel_name, counter_name").fetchall()
for r in rows:
    print('%-28s %-24s %.4g (n=%d)' % (r[0][:28], r[1], r[2], r[3]))
PY
done
