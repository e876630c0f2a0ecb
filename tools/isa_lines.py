"""Where do a kernel's instructions come from?  Static histogram of VALU issue slots per
source function / source line, from an ISA listing compiled with -gline-tables-only:

    cd xrt_amd/csrc && hipcc <flags> -gline-tables-only -c reflect_hot.hip -o /tmp/isa/x.o --save-temps=obj
    python tools/isa_lines.py /tmp/isa/reflect_hot-hip-amdgcn-amd-amdhsa-gfx950.s 'reflect_fused_dcm<xrt::ThickXtal<0>' [--lines]

Quarter-rate fp64 instructions (v_rcp/rsq/sqrt_f64, fp64 conversions) are weighted 4 slots,
v_div_* helpers and 64-bit integer multiplies 1-4 (see RATE)."""
import re
import subprocess
import sys
from collections import Counter, defaultdict

QUARTER = ('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64', 'v_div_scale_f64', 'v_div_fmas_f64',
           'v_div_fixup_f64', 'v_frexp_mant_f64', 'v_ldexp_f64', 'v_trig_preop_f64',
           'v_cvt_f64', 'v_cvt_i32_f64', 'v_cvt_u32_f64', 'v_fract_f64', 'v_rndne_f64',
           'v_floor_f64', 'v_ceil_f64', 'v_trunc_f64', 'v_mul_lo_u32', 'v_mul_hi_u32',
           'v_mul_hi_i32', 'v_mad_u64_u32', 'v_mad_i64_i32')


def slots(op):
    # fp64 transcendental-unit ops issue over 16 cycles; div helpers are full rate
    if op.startswith(('v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64')):
        return 4
    return 1


def main():
    path, want = sys.argv[1], sys.argv[2]
    by_line = '--lines' in sys.argv
    files = {}
    funcs = {}   # file -> sorted [(line, name)] from a crude scan of the source
    cur = None
    hist = Counter()
    ops = defaultdict(Counter)
    total = 0
    demangled = None
    loc = (0, 0)
    with open(path) as f:
        for ln in f:
            if cur is None:
                m = re.match(r'^(_Z\w+):', ln)
                if m:
                    name = subprocess.run(['c++filt', m.group(1)], capture_output=True,
                                          text=True).stdout
                    if want in name:
                        cur = m.group(1)
                        demangled = name.strip()
                m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', ln)
                if m:
                    files[int(m.group(1))] = (m.group(2) + '/' + m.group(3)) if m.group(3) else m.group(2)
                continue
            if '.end_amdhsa_kernel' in ln or ln.startswith('.Lfunc_end'):
                break
            m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', ln)
            if m:
                files[int(m.group(1))] = (m.group(2) + '/' + m.group(3)) if m.group(3) else m.group(2)
                continue
            m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', ln)
            if m:
                loc = (int(m.group(1)), int(m.group(2)))
                continue
            m = re.match(r'\s+(v_\w+)', ln)
            if not m:
                continue
            op = m.group(1)
            w = slots(op)
            total += w
            hist[loc] += w
            ops[loc][op] += 1
    print(demangled[:110] if demangled else 'kernel not found')
    print('static VALU slots:', total)
    srccache = {}

    def func_of(fi, line):
        fn = files.get(fi, '?')
        if fn not in srccache:
            src = []
            for cand in (fn, 'xrt_amd/csrc/' + fn.split('/')[-1]):
                try:
                    src = open(cand).read().split('\n')
                    break
                except OSError:
                    pass
            marks = []
            for k, s in enumerate(src, 1):
                m = re.match(r'^(?:__device__|__global__|static|template|inline).*?(\w+)\s*\(', s)
                if m and not s.startswith('template'):
                    marks.append((k, m.group(1)))
                else:
                    m2 = re.match(r'^\s*(?:__device__|__global__).*?\b(\w+)\s*\(', s)
                    if m2:
                        marks.append((k, m2.group(1)))
            srccache[fn] = marks
        name = '?'
        for k, nm in srccache[fn]:
            if k <= line:
                name = nm
            else:
                break
        return fn.split('/')[-1] + ':' + name

    if by_line:
        for (fi, line), c in hist.most_common(60):
            print('%5d  %s:%d  %s' % (c, func_of(fi, line), line,
                                     ' '.join('%s*%d' % kv for kv in ops[(fi, line)].most_common(4))))
    else:
        agg = Counter()
        for (fi, line), c in hist.items():
            agg[func_of(fi, line)] += c
        for k, c in agg.most_common(50):
            print('%5d  %s' % (c, k))


main()
