#!/bin/bash
# HBM bytes of the plot histogram kernels (FETCH_SIZE / WRITE_SIZE, separate passes, KB units;
# read x2 as calibrated on screen_expose_kernel in profiles/hbm_traffic.json) and their times:
#   bash tools/pmc_hist.sh   -> one row per kernel and launch shape
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=.
cat > /tmp/hist_run.py <<'PY'
import torch
from xrt_amd import workloads, plotter as xrtp, runner
n = 10_000_000
oe = workloads.cfg2_toroid()
beam = workloads.synthetic_rays(n, 42)
for f in beam.array_fields():
    beam.dev(f)
gb, lb = oe.reflect(beam)
import sys
for bins in (int(sys.argv[1]),):
    plot = xrtp.XYCPlot('b', (1,), xrtp.XYCAxis('x', 'mm', bins=bins), xrtp.XYCAxis('y', 'mm', bins=bins),
                        caxis=xrtp.XYCAxis('energy', 'eV', bins=bins))
    for _ in range(4):
        runner.accumulate_plot(plot, {'b': lb})
    torch.cuda.synchronize()
PY
for BINS in 128 256; do
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o p -- python /tmp/hist_run.py $BINS > /tmp/pmc_run.log 2>&1 || tail -3 /tmp/pmc_run.log
done
echo "# $BINS x $BINS bins, 1e7 rays"
BINS=$BINS python - <<'PY'
import sqlite3, glob, json
out = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    db = glob.glob('/tmp/pmc_%s/**/*.db' % C, recursive=True)
    if not db:
        print('no db for', C); continue
    c = sqlite3.connect(db[0])
    q = ("select kernel_name, avg(value), count(*) from counters_collection "
         "where counter_name = ? and (kernel_name like '%plot_hist%' or kernel_name like '%hist_%') group by kernel_name")
    rows = c.execute(q, (C,)).fetchall()
    for name, v, n in rows:
        key = name.split('(')[0][:70]
        out.setdefault(key, {})[C] = (v, n)
tot = {}
for key, d in sorted(out.items()):
    rd = d.get('FETCH_SIZE', (0, 0))[0] * 1024 * 2.0
    wr = d.get('WRITE_SIZE', (0, 0))[0] * 1024
    tot['r'] = tot.get('r', 0) + rd; tot['w'] = tot.get('w', 0) + wr
    print('%-72s read %.1f MB  written %.1f MB  (launches %d)' % (key, rd / 1e6, wr / 1e6, d.get('FETCH_SIZE', (0, 0))[1]))
print('per plot: read %.1f MB + written %.1f MB = %.1f MB (algorithmic 440 MB)' % (tot.get('r', 0) / 1e6, tot.get('w', 0) / 1e6, (tot.get('r', 0) + tot.get('w', 0)) / 1e6))
import os
path = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'profiles', 'hist_traffic.json')
try:
    rec = json.load(open(path))
except Exception:
    rec = {}
rec['bins%s' % os.environ['BINS']] = dict(rays=10000000, read_bytes=tot.get('r', 0), write_bytes=tot.get('w', 0), hbm_bytes_per_plot=tot.get('r', 0) + tot.get('w', 0), note='FETCH_SIZE x 1024 x 2 (gfx950 half-count of coalesced reads, calibrated on screen_expose_kernel in hbm_traffic.json) + WRITE_SIZE x 1024, separate --pmc passes, summed over plot_hist_rays / plot_hist_tiles / plot_hist_reduce')
json.dump(rec, open(path, 'w'), indent=1)
PY
done
