"""Turns rocprofv3's rocpd sqlite outputs into the committed summaries under
profiles/:

  python tools/rocpd_summary.py ROUND trace.db [fetch.db write.db [nrays [fetch_nl.db write_nl.db]]]

  profiles/rROUND_kernel_stats.csv   per-kernel calls / total / average (ns)
  profiles/hbm_traffic.json          per-launch HBM bytes of our kernels from
                                     FETCH_SIZE / WRITE_SIZE (KB), corrected by
                                     the ratio measured on the streaming
                                     calibration kernel (screen_expose: exactly
                                     100 B read + 100 B written per ray), as
                                     MI355X_MICROARCH.md (HBM) prescribes.
                                     fetch.db / write.db: tools/profile_workload.py (the
                                     FULL passes bench.py times, 308 / 416 B per ray);
                                     fetch_nl.db / write_nl.db: the same script with
                                     --nolocal (200 B per ray) -> the *_nolocal entries.
"""
import csv
import json
import os
import sqlite3
import sys

OURS = ('reflect_fused_dcm_scr', 'reflect_fused_dcm_marks', 'reflect_dcm_redo_scr',
        'reflect_fused_gen_scr_plot', 'reflect_fused_scr_plot', 'plot_tail_tiles',
        'reflect_fused_plate2', 'reflect_redo_scr', 'reflect_multi_opt', 'multi_decide_opt',
        'reflect_multi_stats', 'reflect_multi_solve', 'reflect_multi_finish',
        'reflect_fused_gen_scr', 'reflect_fused_scr', 'reflect_decide_opt_gen',
        'reflect_redo_verdict', 'geosource_shine_if_kernel', 'screen_expose_if_kernel',
        'reflect_multi', 'multi_to_global_kernel', 'plot_hist_small', 'reflect_fused_xtal', 'reflect_fused_dcm', 'reflect_dcm_exact', 'reflect_decide_dcm',
        'reflect_decide_opt', 'reflect_exact', 'reflect_fused', 'reflect_init',
        'screen_expose_mark_kernel', 'screen_expose_kernel', 'kirchhoff_stream', 'kirchhoff_scan', 'kirchhoff_pack',
        'kirchhoff_finalize', 'und_imap', 'und_sum', 'und_pack', 'aperture_propagate_kernel',
        'plot_hist_rays', 'plot_hist_tiles', 'plot_hist_reduce', 'geosource_shine_kernel',
        'plot_hist_kernel', 'surface_eval_kernel',
        'beam_to_global_kernel')


def short(name):
    for k in OURS:
        if k in name:
            return k
    return name.split('(')[0][-60:]


def variant(name):
    """short name + its template arguments, e.g. reflect_fused<xrt::Spec<0, 1, 1, true>>"""
    k = short(name)
    i = name.find(k)
    j = i + len(k)
    if i >= 0 and name[j:j + 1] == '<':
        depth = 0
        for e in range(j, len(name)):
            depth += name[e] == '<'
            depth -= name[e] == '>'
            if depth == 0:
                return k + name[j:e + 1]
    return k


def kernel_stats(db):
    """One row per (kernel, launch size): the same kernel is launched on 1e7-ray
    beams by the primary bench and on 2e5-sample waves by the SoftiMAX chain;
    an average over both would match neither."""
    c = sqlite3.connect(db)
    rows = c.execute('select name, grid_x, count(*), sum(duration), avg(duration), '
                     'min(duration), max(duration) from kernels group by name, grid_x '
                     'order by sum(duration) desc').fetchall()
    return [(variant(r[0]),) + tuple(r[1:]) for r in rows]


def counter_per_launch(db, counter):
    """{short kernel name: (average per launch, launches, variant)}. A kernel that
    exists in several instantiations (reflect_fused<Spec, mode>: the optimistic pass,
    the redo that usually returns at once, ...) is represented by the instantiation
    with the largest average - the one that did the work."""
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, avg(value), count(*) from '
                     'counters_collection where counter_name=? group by '
                     'kernel_name', (counter,)).fetchall()
    out = {}
    for name, avg, cnt in rows:
        k = short(name)
        if k not in out or avg > out[k][0]:
            out[k] = (avg, cnt, variant(name))
    return out


def main():
    rnd, trace = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'profiles')
    os.makedirs(out, exist_ok=True)
    stats = kernel_stats(trace) if trace != '-' else []    # ('-': the counter passes alone)
    path = os.path.join(out, 'r%s_kernel_stats.csv' % rnd)
    if stats:
        with open(path, 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'grid_threads', 'calls', 'total_ns', 'avg_ns', 'min_ns',
                        'max_ns'])
            for r in stats:
                w.writerow([r[0], r[1], r[2], int(r[3]), round(r[4], 1), r[5], r[6]])
        print('wrote', path)
    for r in stats[:14]:
        print('  %-28s grid %9d calls %4d  avg %12.1f us' % (r[0], r[1], r[2], r[4] / 1e3))
    if len(sys.argv) >= 5:
        nrays = float(sys.argv[5]) if len(sys.argv) > 5 else 1e7
        fetch = counter_per_launch(sys.argv[3], 'FETCH_SIZE')
        write = counter_per_launch(sys.argv[4], 'WRITE_SIZE')
        cal_r = 100. * nrays / (fetch['screen_expose_kernel'][0] * 1024.)
        cal_w = 100. * nrays / (write['screen_expose_kernel'][0] * 1024.)
        res = {'_calibration': {
            'kernel': 'screen_expose_kernel', 'rays': nrays,
            'known_read_bytes': 100. * nrays, 'known_write_bytes': 100. * nrays,
            'FETCH_SIZE_KB': fetch['screen_expose_kernel'][0],
            'WRITE_SIZE_KB': write['screen_expose_kernel'][0],
            'read_correction': cal_r, 'write_correction': cal_w,
            'note': 'bytes = counter_KB * 1024 * correction; correction = known '
                    'bytes / counted bytes on a pure streaming kernel with the '
                    'same 8 B/lane SoA access pattern'}}
        for k in OURS:
            if k in fetch and k in write:
                rb = fetch[k][0] * 1024. * cal_r
                wb = write[k][0] * 1024. * cal_w
                res[k] = dict(variant=fetch[k][2], FETCH_SIZE_KB=fetch[k][0],
                              WRITE_SIZE_KB=write[k][0],
                              launches=fetch[k][1], read_bytes=rb, write_bytes=wb,
                              hbm_bytes_per_launch=rb + wb)
        if len(sys.argv) >= 8:
            # the passes that leave their local beams out, from a run of their own (the kernel
            # names are the same: out_local == NULL is a run-time switch)
            f_nl = counter_per_launch(sys.argv[6], 'FETCH_SIZE')
            w_nl = counter_per_launch(sys.argv[7], 'WRITE_SIZE')
            c_r = 100. * nrays / (f_nl['screen_expose_kernel'][0] * 1024.)
            c_w = 100. * nrays / (w_nl['screen_expose_kernel'][0] * 1024.)
            for k in ('reflect_fused', 'reflect_fused_dcm'):
                if k in f_nl and k in w_nl:
                    rb = f_nl[k][0] * 1024. * c_r
                    wb = w_nl[k][0] * 1024. * c_w
                    res[k + '_nolocal'] = dict(
                        variant=f_nl[k][2], FETCH_SIZE_KB=f_nl[k][0], WRITE_SIZE_KB=w_nl[k][0],
                        launches=f_nl[k][1], read_bytes=rb, write_bytes=wb,
                        hbm_bytes_per_launch=rb + wb, read_correction=c_r, write_correction=c_w,
                        shape='out_local NULL: 100 B read + 100 B written per ray')
        for k, b in (('reflect_fused', 308.), ('reflect_fused_dcm', 416.)):
            if k in res:
                res[k]['shape'] = 'the full pass bench.py times: %d B per ray' % b
                res[k]['algorithmic_bytes'] = b * nrays
        path = os.path.join(out, 'hbm_traffic.json')
        with open(path, 'w') as f:
            json.dump(res, f, indent=1)
        print('wrote', path)
        for k, v in res.items():
            if not k.startswith('_'):
                print('  %-24s read %.3e B  write %.3e B' % (k, v['read_bytes'], v['write_bytes']))
        print('  calibration read x%.3f write x%.3f' % (cal_r, cal_w))


if __name__ == '__main__':
    main()
