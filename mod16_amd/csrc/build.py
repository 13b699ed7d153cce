#!/usr/bin/env python3
'''Builds the native libraries for gfx950 in-tree; hipcc cross-compiles without a GPU and the
.so files travel with the repo snapshot.

  mod16_amd/libmod16hip.so      the product: one arithmetic, one launch geometry
  mod16_amd/libmod16hip_exp.so  the same sources with -DMOD16_EXPERIMENTS: the launch-geometry
                                overrides (MOD16_NO_DMA, MOD16_RUN_SHIFT, ...) that tools/ and a
                                few tests use to reach the other schedules. Never loaded by the
                                product (mod16_amd._lib.load_experiments() is for tests / tools).

Both carry a build id -- a digest of the sources and flags -- that `mod16_build_id()` returns.'''
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'mod16_capi.hip')
DEPS = [SRC] + sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.hpp')) + [
    os.path.join(os.path.dirname(PKG), 'include', 'mod16_hip.h')]
OUT = os.path.join(PKG, 'libmod16hip.so')
OUT_EXP = os.path.join(PKG, 'libmod16hip_exp.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def build_id(extra=()):
    '''Digest of what determines the code of the library: sources, header, flags, compiler.'''
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS + list(extra)).encode())
    try:
        h.update(subprocess.check_output([HIPCC, '--version'], stderr=subprocess.STDOUT))
    except (OSError, subprocess.CalledProcessError):
        pass
    return h.hexdigest()[:16]


def built_id(out):
    '''The build id inside a built library (None: no file, or one from before the marker).'''
    import re
    try:
        with open(out, 'rb') as f:
            m = re.search(rb'mod16-build-id=([0-9a-f]{16})', f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def up_to_date(out, extra=()):
    '''The library on disk was built from the sources on disk: its id IS their digest (round 5
    compared file dates, and a snapshot that already held a library never met the compiler).'''
    return built_id(out) == build_id(extra)


def command(out, extra):
    name = os.path.basename(out)
    return [HIPCC] + FLAGS + list(extra) + ['-DMOD16_BUILD_ID="%s"' % build_id(extra),
                                            '-Wl,-soname,' + name, '-o', out, SRC]


def build(force=False, verbose=True, experiments=True):
    '''Compiles what is out of date (the two libraries side by side).'''
    todo = [(OUT, [])] + ([(OUT_EXP, ['-DMOD16_EXPERIMENTS'])] if experiments else [])
    procs = []
    for out, extra in todo:
        if not force and up_to_date(out, extra):
            continue
        cmd = command(out, extra)
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if procs:
        write_build_info()
    return OUT


def write_build_info():
    '''mod16_amd/build_info.json: the build id next to the git commit the sources came from (the
    GPU box has no .git); tools/run_profiles.sh copies it into the profile records.'''
    import json
    root = os.path.dirname(PKG)

    def git(*args):
        try:
            return subprocess.check_output(['git', '-C', root] + list(args), stderr=subprocess.DEVNULL).decode().strip()
        except (OSError, subprocess.CalledProcessError):
            return None
    dirty = git('status', '--porcelain', '--', 'mod16_amd/csrc', 'include')
    info = {'build_id': build_id(), 'git_commit': git('rev-parse', 'HEAD'),
            'git_dirty': bool(dirty) if dirty is not None else None}
    with open(os.path.join(PKG, 'build_info.json'), 'w') as f:
        json.dump(info, f, indent=1)


if __name__ == '__main__':
    build(force='--force' in sys.argv)
