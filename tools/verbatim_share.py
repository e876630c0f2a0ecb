"""Development check (build container only: reads /root/reference): for every source
file of the package, the share of its non-trivial lines that also occur verbatim
(whitespace-stripped) somewhere in the reference tree, and the longest run of
consecutive such lines.   python tools/verbatim_share.py [paths...]"""
import os
import sys

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def nontrivial(line):
    t = line.strip()
    if len(t) < 12 or t.startswith(('#', '//', '"""', "'''", 'import ', 'from ', '@')):
        return None
    if t in ('return', 'else:', 'try:', 'pass', 'continue', 'break'):
        return None
    return t


def ref_lines():
    seen = set()
    for d, _, files in os.walk(REF):
        if '/.git' in d:
            continue
        for f in files:
            if f.endswith(('.py', '.cl', '.c', '.h', '.cpp')):
                try:
                    with open(os.path.join(d, f), errors='ignore') as fh:
                        for ln in fh:
                            t = nontrivial(ln)
                            if t:
                                seen.add(t)
                except OSError:
                    pass
    return seen


def main():
    ref = ref_lines()
    verbose = '-v' in sys.argv
    paths = [a for a in sys.argv[1:] if a != '-v']
    if not paths:
        for d, _, files in os.walk(os.path.join(ROOT, 'xrt_amd')):
            for f in files:
                if f.endswith(('.py', '.hip', '.h')):
                    paths.append(os.path.join(d, f))
        paths += [os.path.join(ROOT, f) for f in ('bench.py', '__graft_entry__.py')]
    rows = []
    for p in sorted(paths):
        with open(p, errors='ignore') as fh:
            lines = fh.readlines()
        tot = hit = run = best = 0
        best_at = 0
        for k, ln in enumerate(lines):
            t = nontrivial(ln)
            if t is None:
                continue
            tot += 1
            if t in ref:
                if verbose:
                    print('%s:%d: %s' % (os.path.relpath(p, ROOT), k + 1, t))
                hit += 1
                run += 1
                if run > best:
                    best, best_at = run, k + 1
            else:
                run = 0
        if tot:
            rows.append((hit / tot, hit, tot, best, best_at, os.path.relpath(p, ROOT)))
    for share, hit, tot, best, at, p in sorted(rows, reverse=True):
        print('%5.1f %%  %4d / %4d   longest run %3d (ends line %4d)   %s'
              % (100 * share, hit, tot, best, at, p))


if __name__ == '__main__':
    main()
