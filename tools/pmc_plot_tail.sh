#!/bin/bash
# HBM bytes of ONE run_ray_tracing iteration with the plot in the tail of the pass (round 6) against the
# same iteration with the plot's own launches (fusion of the screen only, XRT_PLOT_TAIL_OFF=1): device source
# -> toroid -> screen -> 256 x 256 XYCPlot, 1e7 rays. FETCH_SIZE / WRITE_SIZE in separate passes, KB units,
# read x2 (calibrated on screen_expose_kernel, profiles/hbm_traffic.json).  -> profiles/plot_tail_traffic.json
#   bash tools/pmc_plot_tail.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
cat > /tmp/tail_run.py <<'PY'
import sys
import torch
from xrt_amd import workloads, runner
from xrt_amd.backends.raycing import run as rr
bl, run_process, make_plot = workloads.e2e_beamline(10_000_000)
rr.run_process = run_process
runner.run_ray_tracing([make_plot()], repeats=5, beamLine=bl)
torch.cuda.synchronize()
PY
for MODE in tail separate; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${MODE}_$C
  if [ $MODE = separate ]; then export XRT_PLOT_TAIL_OFF=1; else unset XRT_PLOT_TAIL_OFF; fi
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_${MODE}_$C -o p -- python /tmp/tail_run.py > /tmp/pmc_run.log 2>&1 || tail -3 /tmp/pmc_run.log
done
done
unset XRT_PLOT_TAIL_OFF
python - <<'PY'
import sqlite3, glob, json, os
rec = {}
for mode in ('tail', 'separate'):
    out = {}
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        db = glob.glob('/tmp/pmc_%s_%s/**/*.db' % (mode, C), recursive=True)
        if not db:
            print('no db for', mode, C); continue
        c = sqlite3.connect(db[0])
        q = ("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and "
             "(kernel_name like '%plot_%' or kernel_name like '%gen_scr%' or kernel_name like '%redo_scr%') "
             "group by kernel_name")
        for name, v, n in c.execute(q, (C,)).fetchall():
            out.setdefault(name.split('(')[0].replace('void ', '')[:64], {})[C] = (v, n)
    print('# %s' % ('the plot in the tail of the pass' if mode == 'tail' else 'the plot as launches of its own'))
    tot_r = tot_w = 0.
    rows = {}
    for key, d in sorted(out.items()):
        rd = d.get('FETCH_SIZE', (0, 0))[0] * 1024 * 2.0
        wr = d.get('WRITE_SIZE', (0, 0))[0] * 1024
        tot_r += rd; tot_w += wr
        rows[key] = dict(read_bytes=rd, write_bytes=wr)
        print('%-66s read %7.1f MB  written %7.1f MB  (launches %d)' % (key, rd / 1e6, wr / 1e6, d.get('FETCH_SIZE', (0, 0))[1]))
    print('per iteration: read %.1f MB + written %.1f MB = %.1f MB' % (tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6))
    rec[mode] = dict(rays=10000000, kernels=rows, read_bytes=tot_r, write_bytes=tot_w, hbm_bytes_per_iteration=tot_r + tot_w)
rec['note'] = ('one run_ray_tracing iteration, device source -> toroid -> screen -> 256 x 256 XYCPlot, 1e7 rays: the pass and '
               'the plot kernels; FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, separate --pmc passes (tools/pmc_plot_tail.sh). '
               'Algorithmic: tail = 20 B record + 0.5 B table row per ray written and read back (410 MB); separate = 100 B '
               'image written + 44 B read + 20 B records written and read (1840 MB)')
json.dump(rec, open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'profiles', 'plot_tail_traffic.json'), 'w'), indent=1)
PY
