#!/usr/bin/env python3
"""Weighted VALU issue-cycle estimate of an instruction range of one kernel in a hipcc -save-temps gfx950 .s file.

Issue costs per wave64 instruction per SIMD measured on MI355X with the shader clock read in-kernel
(tools/probes/valu_rate_probe{,2,3}.hip, profiles/r02_valu_issue_rates.txt):
  2.45 cycles  v_add/sub (u32, f32), v_and/or/xor, v_lshrrev, v_ashrrev, v_mov, v_mul/fma/fmac/fmamk f32
               -- only when every source is a VGPR, an inline constant or a 32-bit literal;
  4.3  cycles  the same opcodes with an SGPR source operand, and every other VALU opcode (v_lshlrev_b32 included);
  8.4  cycles  v_mfma_f32_4x4x1 (does not overlap with VALU issue of the other waves).
usage: isa_cycles.py <file.s> <kernel-substring> [first_line last_line]   (line numbers inside the kernel body)"""
import collections
import re
import sys

FAST = re.compile(r'^v_(add|sub|subrev)_(u32|f32)|^v_(and|or|xor|not)_b32|^v_(lshrrev|ashrrev)_(b32|i32)'
                  r'|^v_(fma|fmac|fmamk|fmaak|mul|mac)_f32|^v_mov_b32')
SGPR_SRC = re.compile(r'(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec|ttmp\d+|m0)(?![a-z0-9_])')


def cost(line):
    t = line.split(None, 1)
    op = t[0]
    if not op.startswith('v_'):
        return 0.0
    if op.startswith('v_mfma'):
        return 8.4
    if not FAST.match(op):
        return 4.3
    srcs = t[1].split(',', 1)[1] if len(t) > 1 and ',' in t[1] else ''
    return 4.3 if SGPR_SRC.search(srcs) else 2.45


def kernel_lines(path, pat):
    s = open(path).read()
    for f in re.split(r'\n\t\.globl\t', s)[1:]:
        name = f.split('\n', 1)[0].strip()
        if pat in name:
            return name, f.split('.end_amdhsa_kernel')[0].split('\n')
    raise SystemExit('kernel not found')


def main():
    name, lines = kernel_lines(sys.argv[1], sys.argv[2])
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(lines)
    cyc = collections.Counter()
    cnt = collections.Counter()
    for l in lines[lo:hi]:
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;':
            continue
        op = t.split()[0]
        c = cost(t)
        key = op + (' [sgpr src]' if c == 4.3 and FAST.match(op) else '')
        cnt[key] += 1
        cyc[key] += c
    n = sum(cnt.values())
    valu = sum(v for k, v in cnt.items() if k.startswith('v_'))
    print(name[:80], 'lines', lo, hi, 'instructions', n, 'VALU', valu, 'LDS', sum(v for k, v in cnt.items() if k.startswith('ds_')),
          'SALU', sum(v for k, v in cnt.items() if k.startswith('s_')), 'est. VALU cycles %.0f' % sum(cyc.values()))
    for k, v in sorted(cyc.items(), key=lambda kv: -kv[1])[:45]:
        if v:
            print('   %-36s n=%-4d cycles=%.0f' % (k, cnt[k], v))


if __name__ == '__main__':
    main()
