#!/usr/bin/env python3
"""Register use of the gfx950 kernels in a hipcc -S listing: VGPRs, spills, scratch,
accumulator-file parking, instruction counts. Usage: tools/kernel_regs.py file.s [substring]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.strip().split('\n')
    except OSError:
        return names


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    meta = re.findall(r'\.name:\s+(\S+)\n((?:.*\n){0,40}?)\s*\.wavefront_size', text)
    rows = []
    for name, block in meta:
        get = lambda key: int(re.search(r'\.%s:\s*(\d+)' % key, block).group(1)) if re.search(r'\.%s:\s*(\d+)' % key, block) else -1
        rows.append((name, get('vgpr_count'), get('sgpr_count'), get('vgpr_spill_count'),
                     get('private_segment_fixed_size'), get('group_segment_fixed_size')))
    bodies = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', text)
    ninstr = {}
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = body.split('.section')[0]
        ins = [l.strip().split()[0] for l in body.split('\n')
               if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;']
        ninstr[name] = (len(ins), sum(i.startswith('scratch_') for i in ins),
                        sum(i.startswith('v_accvgpr') for i in ins))
    names = demangle([r[0] for r in rows])
    for (name, v, s, sp, priv, lds), nice in zip(rows, names):
        if pat and pat not in nice:
            continue
        n, scr, acc = ninstr.get(name, (0, 0, 0))
        print('vgpr %3d sgpr %3d spill %d scratch %d(%d B) acc %d lds %6d instr %5d  %s'
              % (v, s, sp, scr, priv, acc, lds, n, nice[:140]))


if __name__ == '__main__':
    main()
