"""Rewrites the kernel times DESIGN.md quotes from the committed rocprofv3 summary
(profiles/rNN_kernel_stats.csv), so that the document cannot drift from the profile:
    python tools/update_design_numbers.py 03
(tests/test_docs_match_profiles.py checks the same figures.)"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else '03'
rows = list(csv.reader(open(os.path.join(ROOT, 'profiles', 'r%s_kernel_stats.csv' % rnd))))[1:]


def one(pred):
    hits = [(int(r[2]), float(r[4])) for r in rows if pred(r[0], int(r[1]), int(r[2]), float(r[4]))]
    assert len(hits) == 1, hits
    return hits[0]


cfg2 = one(lambda k, g, n, a: k.startswith('reflect_fused<xrt::Spec<0, 1, 1, true>, 0>') and g >= 10_000_000)
dcm = one(lambda k, g, n, a: k.startswith('reflect_fused_dcm<xrt::ThickXtal<0>') and g >= 10_000_000)
k4 = one(lambda k, g, n, a: k == 'kirchhoff_stream<4>' and n <= 12 and a > 2e8)
kg = one(lambda k, g, n, a: k == 'kirchhoff_stream<4>' and g == 4515840)
import json
und_ms = json.load(open(os.path.join(ROOT, 'profiles', 'r%s_bench_under_rocprof.json' % rnd)))['undulator']['ms']
path = os.path.join(ROOT, 'DESIGN.md')
text = open(path).read()
subs = (
    (r'(rocprofv3 average over )\d+( launches \*\*)[\d.]+( µs\*\*)',
     r'\g<1>%d\g<2>%.1f\g<3>' % (cfg2[0], cfg2[1] * 1e-3)),
    (r'(kernel \*\*)[\d.]+( µs\*\* \(rocprofv3, )\d+( launches\))',
     r'\g<1>%.1f\g<2>%d\g<3>' % (dcm[1] * 1e-3, dcm[0])),
    (r'(rocprofv3 average )[\d.]+( ms over 6 launches, HIP events in the same run)',
     r'\g<1>%.1f\g<2>' % (k4[1] * 1e-6)),
    (r'(grid 4515840, )[\d.]+( ms over )\d+( launches)',
     r'\g<1>%.2f\g<2>%d\g<3>' % (kg[1] * 1e-6, kg[0])),
    (r'(`und_imap` 2\^20 rays × 48 nodes \*\*)[\d.]+( ms in\s+the bench)', r'\g<1>%.3f\g<2>' % und_ms),
)
for pat, rep in subs:
    text, n = re.subn(pat, rep, text)
    assert n == 1, pat
open(path, 'w').write(text)
print('cfg2 %.1f us (%d), dcm %.1f us (%d), cfg4 %.1f ms, general %.2f ms, und_imap %.1f us' % (
    cfg2[1] * 1e-3, cfg2[0], dcm[1] * 1e-3, dcm[0], k4[1] * 1e-6, kg[1] * 1e-6, und_ms * 1e3))
