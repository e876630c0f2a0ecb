#!/bin/bash
# Where the GPU idles in a SoftiMAX run: kernel trace of tools/probe_softimax_gaps.py, gaps between
# consecutive kernels of the last run, with the kernels on either side.
#   gpurun -- 'bash tools/softimax_gaps.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/sg.py <<'PY'
import types, numpy as np, torch
import xrt_amd.backends.raycing as raycing
import xrt_amd.backends.raycing.sources as rs
import xrt_amd.backends.raycing.apertures as ra
import xrt_amd.backends.raycing.oes as roe
import xrt_amd.backends.raycing.materials as rm
import xrt_amd.backends.raycing.screens as rsc
import xrt_amd.backends.raycing.waves as rw
from xrt_amd.workloads import SoftiMAX
mods = types.SimpleNamespace(raycing=raycing, rs=rs, ra=ra, roe=roe, rm=rm, rsc=rsc, rw=rw)
np.random.seed(1)
scene = SoftiMAX(mods, nrays=200000)
for _ in range(3):
    scene.run()
    torch.cuda.synchronize()
PY
rm -rf /tmp/sg
PYTHONPATH=. rocprofv3 --kernel-trace -d /tmp/sg -o sg -- python /tmp/sg.py > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/sg/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the last run = the last third of the big integrals
big = [i for i, r in enumerate(rows) if 'kirchhoff_stream' in r[0] and r[2] - r[1] > 2e7]
first = big[-7]
t0 = rows[first][1]
seg = rows[first - 400 if first > 400 else 0:]
# find start of last run: first kernel after the previous run's last big integral
prev_big = big[-8] if len(big) >= 8 else 0
seg = rows[prev_big + 1:]
idle = 0.
busy = 0.
gaps = []
for a, b in zip(seg, seg[1:]):
    g = b[1] - a[2]
    if g > 0:
        idle += g
        gaps.append((g, a[0][:40], b[0][:40]))
    busy += a[2] - a[1]
print('last run window: busy %.1f ms, idle %.1f ms, kernels %d' % (busy / 1e6, idle / 1e6, len(seg)))
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print('%7.2f ms  after %-40s before %s' % (g / 1e6, a, b))
from collections import Counter
cnt = Counter(r[0].split('(')[0][-48:] for r in seg)
tim = Counter()
for r in seg:
    tim[r[0].split('(')[0][-48:]] += r[2] - r[1]
for name, k in cnt.most_common(16):
    print('%5d x %-50s %8.2f ms' % (k, name, tim[name] / 1e6))
PY
