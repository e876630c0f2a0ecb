"""Per-kernel table from the rocpd database that `rocprofv3 --kernel-trace --stats -d DIR` left
under DIR: name, grid, calls, average us, share. Usage: python tools/prof_stats.py DIR [rows]."""
import glob
import sqlite3
import sys

found = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)
if not found:
    sys.exit('no rocpd database under ' + sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = sqlite3.connect(found[0]).execute(
    'select name, grid_x, count(*), avg(duration), sum(duration) from kernels '
    'group by name, grid_x order by sum(duration) desc').fetchall()
total = sum(r[4] for r in rows)
for r in rows[:top]:
    print('%-72s grid %9d  n %4d  avg %8.1f us  %5.1f %%'
          % (r[0].split('(')[0][:72], r[1], r[2], r[3] / 1e3, 100. * r[4] / total))
