"""Instruction audit of an inner loop from hipcc's --save-temps ISA: every
instruction of the loop body in a bucket -- fp64 arithmetic by opcode (what the numpy expression
mandates), and what is NOT mandated by the arithmetic: moves / selects, integer and address math,
conversions, LDS and memory instructions, scalar instructions, waits.

    python tools/kisa_audit.py FILE.s KERNEL_SUBSTRING [LABEL_SUBSTRING] [--per N] [--min M]

prints one table per basic block of the kernel that holds at least M (default 100) instructions
and branches back to itself or is the largest; --per N divides the counts by N (pairs or nodes
the block handles per lane)."""
import re
import sys
from collections import Counter


def blocks_of(text, kernel):
    i = text.index(kernel)
    i = text.index(':\n', i)
    j = text.index('.end_amdhsa_kernel', i)
    blocks, cur = [], None
    for ln in text[i:j].split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', ln)
        if m:
            tag = re.search(r'(GenKern|FastKern)I\w+?EE', m.group(2))
            cur = [m.group(1), [], tag.group(0) if tag else '']
            blocks.append(cur)
        elif cur is not None:
            t = ln.strip()
            if t and not t.startswith(('.', ';', '//')):
                cur[1].append(t)
    return blocks


F64 = ('v_fmac_f64', 'v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_rsq_f64', 'v_rcp_f64', 'v_sqrt_f64', 'v_div_',
       'v_fract_f64', 'v_floor_f64', 'v_rndne_f64', 'v_trunc_f64', 'v_ldexp_f64', 'v_frexp_',
       'v_max_f64', 'v_min_f64', 'v_trig_preop_f64', 'v_ceil_f64')


def bucket(op):
    if op.startswith(F64):
        return 'fp64 arithmetic', op.split('_e')[0]
    if op.startswith('v_cmp') or op.startswith('v_cndmask') or op.startswith('v_mov') or \
            op.startswith('v_accvgpr') or op.startswith('v_readlane') or \
            op.startswith('v_writelane') or op.startswith('v_readfirstlane') or \
            op.startswith('v_bfi') or op.startswith('v_perm') or op.startswith('v_swap'):
        return 'moves / selects / compares', op.split('_e')[0]
    if op.startswith('v_cvt'):
        return 'conversions', op.split('_e')[0]
    if op.startswith('v_'):
        return 'integer / address / bit math (VALU)', op.split('_e')[0]
    if op.startswith('ds_'):
        return 'LDS', op
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'memory', op
    if op.startswith('s_waitcnt') or op.startswith('s_nop') or op.startswith('s_sleep'):
        return 'waits', op
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'scalar loads', op
    if op.startswith('s_'):
        return 'scalar ALU / branches', op.split('_e')[0]
    return 'other', op


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    per = float(sys.argv[sys.argv.index('--per') + 1]) if '--per' in sys.argv else 1.
    least = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 100
    text = open(args[0]).read()
    want = args[2] if len(args) > 2 else None
    for name, body, tag in blocks_of(text, args[1]):
        ins = [b.split()[0] for b in body]
        loops = any(name in b for b in body if b.startswith(('s_cbranch', 's_branch')))
        if want:
            if want != name and want not in tag:
                continue
        elif len(ins) < least or not loops:
            continue
        table = {}
        for op in ins:
            cls, key = bucket(op)
            table.setdefault(cls, Counter())[key] += 1
        valu = sum(1 for op in ins if op.startswith('v_'))
        print('== %s %s %s: %d instructions, %d VALU (%.2f per unit)' % (args[1], name, tag, len(ins),
                                                                        valu, valu / per))
        for cls in ('fp64 arithmetic', 'moves / selects / compares',
                    'integer / address / bit math (VALU)', 'conversions', 'LDS', 'memory',
                    'scalar loads', 'scalar ALU / branches', 'waits', 'other'):
            c = table.get(cls)
            if not c:
                continue
            n = sum(c.values())
            print('  %-38s %4d  (%.2f per unit)  %s' % (
                cls, n, n / per, ', '.join('%s %d' % kv for kv in c.most_common(8))))


if __name__ == '__main__':
    main()
