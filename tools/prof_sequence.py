"""The kernels of a rocpd database in launch order, the last N of them (name, grid, us):
   python tools/prof_sequence.py DIR [N]."""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sqlite3.connect(db).execute(
    'select name, grid_x, duration, start from kernels order by start').fetchall()
t0 = rows[-n][3]
for name, grid, dur, start in rows[-n:]:
    print('%9.1f us  %-70s grid %9d  %8.1f us' % ((start - t0) / 1e3, name.split('(')[0][:70],
                                                 grid, dur / 1e3))
