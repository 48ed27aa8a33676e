#!/usr/bin/env python3
"""Static instruction histogram per kernel from a hipcc -save-temps gfx950 .s file.
usage: isa_count.py <file.s> [name-substring ...]"""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    funcs = re.split(r'\n\t\.globl\t', s)
    for f in funcs[1:]:
        name = f.split('\n', 1)[0].strip()
        if pats and not any(p in name for p in pats):
            continue
        body = f.split('.end_amdhsa_kernel')[0]
        ins = []
        for l in body.split('\n'):
            t = l.strip()
            if not l.startswith('\t') or not t or t[0] in '.;':
                continue
            ins.append(t.split()[0])
        c = collections.Counter()
        for i in ins:
            if i.startswith('v_'):
                c['valu'] += 1
            elif i.startswith('s_'):
                c['salu'] += 1
            elif i.startswith('ds_'):
                c['lds'] += 1
            elif i.startswith(('global_load', 'buffer_load', 'flat_load')):
                c['vmem_load'] += 1
            elif i.startswith(('global_store', 'buffer_store', 'flat_store')):
                c['vmem_store'] += 1
            else:
                c['other'] += 1
        top = collections.Counter(ins).most_common(14)
        print(name[:90])
        print('   total %d  %s' % (len(ins), dict(c)))
        print('   top: ' + ', '.join('%s:%d' % kv for kv in top))


if __name__ == '__main__':
    main()
