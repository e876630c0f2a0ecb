cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in ${LIBS:-"" xrt_amd/ab/libxrt_nohot.so}; do
  echo "=== ${L:-default}"
  for B in ${BINS:-128 256 512}; do
    rm -rf /tmp/hp
    XRT_HIP_LIBRARY=${L/default/} rocprofv3 --kernel-trace --stats -d /tmp/hp -o hp -- env PYTHONPATH=. python tools/probe_hist.py $B > /dev/null 2>&1
    python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/hp/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
print($B, ' '.join("%s %.0f" % (r[0].split('(')[0][-16:], r[1] / 1e3) for r in c.execute("select name, avg(duration) from kernels where name like '%plot_hist%' group by name order by name")))
PY
  done
  XRT_HIP_LIBRARY=${L/default/} PYTHONPATH=. XRT_E2E_NO_HOST=1 python -c "
import bench, json
r = bench.bench_e2e(10_000_000, repeats=10)
print('e2e ms', round(r['ms_per_iteration'],3), {k: round(v,3) for k,v in r['gpu_ms_by_step'].items()})"
done
