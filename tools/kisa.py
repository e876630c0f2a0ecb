"""Per-loop instruction summary of the Kirchhoff kernels from hipcc's --save-temps ISA:
    cd xrt_amd/csrc && hipcc <flags> -c kirchhoff.hip -o build/kirchhoff.o --save-temps=obj
    python tools/kisa.py [ppt]"""
import re
import sys
from collections import Counter
import os
path = os.environ.get('KISA', '/tmp/kisa/kirchhoff-hip-amdgcn-amd-amdhsa-gfx950.s')
s = open(path).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', s):
    n = m.group(1)
    if 'stream' in n:
        print(n[:40], 'scratch %s sgpr %s (spill %s) vgpr %s (spill %s)' % m.groups()[1:])
ppt = sys.argv[1] if len(sys.argv) > 1 else '2'
i = s.index('_ZN3xrt16kirchhoff_streamILi%sEEEv' % ppt)
i = s.index(':\n', i)
j = s.index('.end_amdhsa_kernel', i)
blocks = []
cur = None
for ln in s[i:j].split('\n'):
    if re.match(r'^\.LBB\d+_\d+:', ln):
        cur = [ln, []]
        blocks.append(cur)
    elif cur is not None:
        cur[1].append(ln.strip())
for name, body in blocks:
    ins = [b.split()[0] for b in body if b and not b.startswith(('.', ';', '//'))]
    nload = sum(1 for x in ins if x.startswith('s_load_dwordx16'))
    if nload >= 2 and len(ins) > 120 and 'stream_step' in name:
        c = Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        f64 = sum(v for k, v in c.items() if 'f64' in k)
        steps = 3 if nload == 3 else 2
        tag = re.search(r'(GenKern|FastKern)I([A-Za-z0-9]+?)EE', name)
        print('%-28s steps %d  valu/pair %.2f  f64/pair %.2f  lanes r/w %d/%d  rsq %d' % (
            tag.group(0) if tag else name[:28], steps, valu / steps / int(ppt),
            f64 / steps / int(ppt), c.get('v_readlane_b32', 0), c.get('v_writelane_b32', 0),
            c.get('v_rsq_f64_e32', 0)))
