#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in a hipcc -S listing.
Usage: tools/isa_count.py file.s [name-substring] [--top N]"""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else ''
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 0
    text = open(path).read()
    parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', text)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        if pat not in name:
            continue
        body = body.split('.section')[0]
        ins = []
        for line in body.split('\n'):
            t = line.strip()
            if not line.startswith('\t') or not t or t[0] in '.;':
                continue
            ins.append(t.split()[0])
        c = collections.Counter(ins)
        f64 = sum(v for k, v in c.items() if '_f64' in k)
        trans = sum(v for k, v in c.items() if re.match(r'v_(rcp|rsq|sqrt|exp|log)_', k))
        vmem = sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))
        lds = sum(v for k, v in c.items() if k.startswith('ds_'))
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        salu = sum(v for k, v in c.items() if k.startswith('s_'))
        print('%s\n   total %d  valu %d (f64 %d, transcendental %d)  salu %d  vmem %d  lds %d'
              % (name, len(ins), valu, f64, trans, salu, vmem, lds))
        for k, v in c.most_common(top):
            print('      %-28s %d' % (k, v))


if __name__ == '__main__':
    main()
