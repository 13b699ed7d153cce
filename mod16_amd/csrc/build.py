#!/usr/bin/env python3
'''Builds libmod16hip.so for gfx950 in-tree (mod16_amd/libmod16hip.so).
hipcc cross-compiles without a GPU; the .so travels with the repo snapshot.'''
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'mod16_capi.hip')
DEPS = [SRC] + sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.hpp')) + [
    os.path.join(os.path.dirname(PKG), 'include', 'mod16_hip.h')]
OUT = os.path.join(PKG, 'libmod16hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS + [__file__])


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    cmd = [HIPCC] + FLAGS + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
