"""CPU: the kernel times DESIGN.md quotes for the current round are the ones in the committed
rocprofv3 summary (profiles/r06_kernel_stats.csv) -- the documents drifted from the profiles
once."""
import csv
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def averages():
    """{(kernel variant, grid): (calls, average ns)}"""
    out = {}
    with open(os.path.join(ROOT, 'profiles', 'r06_kernel_stats.csv')) as f:
        rows = list(csv.reader(f))
    for r in rows[1:]:
        out[(r[0], int(r[1]))] = (int(r[2]), float(r[4]))
    return out


def quoted(text, pattern):
    m = re.search(pattern, text)
    assert m, pattern
    return float(m.group(1))


def test_design_quotes_the_committed_profile():
    avg = averages()
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    cfg2 = [v for (k, g), v in avg.items()
            if k.startswith('reflect_fused<xrt::Spec<0, 1, 1, true>, 0>') and g >= 10_000_000]
    dcm = [v for (k, g), v in avg.items()
           if k.startswith('reflect_fused_dcm<xrt::ThickXtal<0>') and g >= 10_000_000]
    k4 = [v for (k, g), v in avg.items() if k == 'kirchhoff_stream<4>' and v[0] <= 12 and
          v[1] > 2e8]
    kg = [v for (k, g), v in avg.items() if k == 'kirchhoff_stream<4>' and g == 4515840]
    assert len(cfg2) == len(dcm) == len(k4) == len(kg) == 1
    checks = (
        (cfg2[0], r'rocprofv3 average over (\d+) launches \*\*([\d.]+) µs\*\*', 1e-3),
        (dcm[0], r'kernel \*\*([\d.]+) µs\*\* \(rocprofv3, (\d+) launches\)', 1e-3),
    )
    m = re.search(checks[0][1], text)
    assert m and int(m.group(1)) == cfg2[0][0] and abs(float(m.group(2)) - cfg2[0][1] * 1e-3) < 0.06
    m = re.search(checks[1][1], text)
    assert m and int(m.group(2)) == dcm[0][0] and abs(float(m.group(1)) - dcm[0][1] * 1e-3) < 0.06
    ms = quoted(text, r'rocprofv3 average ([\d.]+) ms over 6 launches, HIP events in the same run')
    assert abs(ms - k4[0][1] * 1e-6) < 0.06
    ms = quoted(text, r'grid 4515840, ([\d.]+) ms over (?:\d+) launches')
    assert abs(ms - kg[0][1] * 1e-6) < 0.006
    # (the map kernel is a persistent launch: its grid no longer tells the 2^20-ray call from
    # the others, so the document quotes the bench line of the profiled run)
    import json
    with open(os.path.join(ROOT, 'profiles', 'r06_bench_under_rocprof.json')) as f:
        und_ms = json.load(f)['undulator']['ms']
    ms = quoted(text, r'`und_imap` 2\^20 rays × 48 nodes \*\*([\d.]+) ms in\s+the bench')
    assert abs(ms - und_ms) < 0.0006


def test_profile_index_is_whole():
    """profiles/README.md is the index that says which file backs which claim: it stays a
    short table and names every file next to it (a refresh once blew it up to 17k lines,
    VERDICT r3 weak #3)."""
    pdir = os.path.join(ROOT, 'profiles')
    text = open(os.path.join(pdir, 'README.md')).read()
    assert len(text.splitlines()) < 200 and len(text) < 60_000
    missing = [f for f in sorted(os.listdir(pdir)) if f != 'README.md' and f not in text]
    assert not missing, missing


def test_counter_traffic_belongs_to_the_timed_launch():
    """profiles/hbm_traffic.json's figures under the headline are those of the launches
    bench.py times -- the full passes, 308 B (cfg2) and 416 B (cfg3) per ray -- not of the
    200-B forms that leave the local beams out (the file once held the
    latter while the bench line and this document quoted it for the former); the 200-B forms
    have entries of their own; DESIGN.md quotes the file's GB; bench.py hands out a figure only
    for a launch of the matching shape."""
    import importlib.util
    import json
    with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as f:
        t = json.load(f)
    n = t['_calibration']['rays']
    for kernel, per_ray in (('reflect_fused', 308.), ('reflect_fused_dcm', 416.),
                            ('reflect_fused_nolocal', 200.), ('reflect_fused_dcm_nolocal', 200.)):
        ratio = t[kernel]['hbm_bytes_per_launch'] / (per_ray * n)
        assert 0.97 <= ratio <= 1.15, (kernel, ratio)
    text = open(os.path.join(ROOT, 'DESIGN.md')).read()
    gb = quoted(text, r'PMC ([\d.]+) GB moved \(`hbm_traffic.json`\)')
    assert abs(gb - t['reflect_fused']['hbm_bytes_per_launch'] / 1e9) < 0.006
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.load_traffic('reflect_fused', 308. * n) == t['reflect_fused']['hbm_bytes_per_launch']
    assert bench.load_traffic('reflect_fused_nolocal', 308. * n) is None     # another shape
    assert bench.load_traffic('reflect_fused', 200. * n) is None
