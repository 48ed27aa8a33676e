#!/usr/bin/env python3
"""Weighted VALU issue-cycle estimate of an instruction range of one kernel in a hipcc -save-temps
gfx950 .s file, using the per-wave64 issue costs measured by tools/probes/valu_rate_probe.hip
(2 cycles: 32-bit add/sub/logic/shift and fp32 add/mul/fma; 4 cycles: everything else).
usage: isa_cycles.py <file.s> <kernel-substring> [first last]"""
import collections
import re
import sys

FAST = re.compile(r'^v_(add|sub|subrev)_(u32|f32|co_u32)|^v_(and|or|xor|not)_b32|^v_(lshlrev|lshrrev|ashrrev)_(b32|i32)'
                  r'|^v_(fma|fmac|fmamk|fmaak|mul|mac)_f32|^v_mov_b32|^v_cndmask_b32')


def cost(op):
    if not op.startswith('v_'):
        return 0
    return 2 if FAST.match(op) else 4


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    for f in re.split(r'\n\t\.globl\t', s)[1:]:
        name = f.split('\n', 1)[0].strip()
        if pat not in name:
            continue
        ins = []
        for l in f.split('.end_amdhsa_kernel')[0].split('\n'):
            t = l.strip()
            if not l.startswith('\t') or not t or t[0] in '.;':
                continue
            ins.append(t.split()[0])
        lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(ins)
        seg = ins[lo:hi]
        c = collections.Counter(seg)
        tot = sum(cost(k) * v for k, v in c.items())
        print(name[:80], 'instructions', len(seg), 'VALU', sum(v for k, v in c.items() if k.startswith('v_')),
              'LDS', sum(v for k, v in c.items() if k.startswith('ds_')), 'est. VALU cycles', tot)
        for k, v in sorted(c.items(), key=lambda kv: -cost(kv[0]) * kv[1])[:40]:
            print('   %-28s n=%-4d cycles=%d' % (k, v, cost(k) * v))
        return


if __name__ == '__main__':
    main()
