#!/usr/bin/env python3
"""Static instruction counts of the LOOPS of a gfx950 kernel in a hipcc -S listing (back edges to
an earlier label): instructions, VALU, float64, transcendental per loop body.
Usage: tools/isa_loops.py file.s mangled-name-substring [...]"""
import collections
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', text)
    for i in range(1, len(parts), 2):
        name = parts[i]
        if not any(p in name for p in sys.argv[2:]):
            continue
        lines = parts[i + 1].split('.section')[0].split('\n')
        labels = {}
        for k, l in enumerate(lines):
            m = re.match(r'(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = k
        print(name)
        for k, l in enumerate(lines):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                a = labels[m.group(1)]
                ins = [x.strip().split()[0] for x in lines[a:k + 1]
                       if x.startswith('\t') and x.strip() and x.strip()[0] not in '.;']
                c = collections.Counter(ins)
                print('   loop at lines %d-%d: %d instructions, %d VALU (%d float64, %d transcendental), %d SALU'
                      % (a, k, len(ins), sum(n for q, n in c.items() if q.startswith('v_')),
                         sum(n for q, n in c.items() if '_f64' in q),
                         sum(n for q, n in c.items() if re.match(r'v_(rcp|rsq|sqrt|exp|log)_', q)),
                         sum(n for q, n in c.items() if q.startswith('s_'))))


if __name__ == '__main__':
    main()
