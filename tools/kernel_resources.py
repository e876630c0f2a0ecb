"""Registers and scratch of every kernel in the built library, read from the code objects
embedded in xrt_amd/libxrt_hip.so (clang offload bundles -> llvm-readelf --notes):
    python tools/kernel_resources.py [substring]
Scratch matters beyond its own traffic: a kernel with a large private segment costs ~12 us of
extra launch overhead per dispatch, also when it returns at once."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'xrt_amd', 'libxrt_hip.so')
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def kernels(lib=LIB):
    """{mangled kernel name: dict(scratch, sgpr, sgpr_spill, vgpr, vgpr_spill, kernarg)}"""
    data = open(lib, 'rb').read()
    out = {}
    for m in re.finditer(MAGIC, data):
        p = m.start()
        count = struct.unpack_from('<Q', data, p + len(MAGIC))[0]
        off = p + len(MAGIC) + 8
        for _ in range(count):
            start, size, tlen = struct.unpack_from('<QQQ', data, off)
            off += 24
            triple = data[off:off + tlen].decode()
            off += tlen
            if 'gfx950' not in triple or not size:
                continue
            with tempfile.NamedTemporaryFile(suffix='.o') as f:
                f.write(data[p + start:p + start + size])
                f.flush()
                notes = subprocess.run([READELF, '--notes', f.name], capture_output=True,
                                       text=True).stdout
            for blk in re.split(r'\n  - \.agpr_count', notes)[1:]:
                def field(key):
                    return int(re.search(r'\.' + key + r':\s+(\d+)', blk).group(1))
                out[re.search(r'\.name:\s+(\S+)', blk).group(1)] = dict(
                    scratch=field('private_segment_fixed_size'), sgpr=field('sgpr_count'),
                    sgpr_spill=field('sgpr_spill_count'), vgpr=field('vgpr_count'),
                    vgpr_spill=field('vgpr_spill_count'), kernarg=field('kernarg_segment_size'))
    return out


if __name__ == '__main__':
    want = sys.argv[1] if len(sys.argv) > 1 else ''
    for name, r in sorted(kernels().items()):
        if want in name:
            print('%-100s scratch %5d  vgpr %3d (spill %d)  sgpr spill %4d  kernarg %d' % (
                name[:100], r['scratch'], r['vgpr'], r['vgpr_spill'], r['sgpr_spill'],
                r['kernarg']))
