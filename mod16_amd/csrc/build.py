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
SRC = os.path.join(HERE, 'mod16_capi.hip')          # all families as ONE unit (variants, listings, tests/host_asan)
CAPI = os.path.join(HERE, 'capi')
FAMILIES = sorted(os.path.join(CAPI, f) for f in os.listdir(CAPI) if f.endswith('.hip'))
DEPS = [SRC] + FAMILIES + sorted(os.path.join(d, f) for d in (HERE, CAPI) for f in os.listdir(d) if f.endswith('.hpp')) + [
    os.path.join(os.path.dirname(PKG), 'include', 'mod16_hip.h')]
OUT = os.path.join(PKG, 'libmod16hip.so')
OUT_EXP = os.path.join(PKG, 'libmod16hip_exp.so')
OBJ = os.path.join(HERE, 'build')                    # object files of the families (git-ignored)
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
COMPILE = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-Wall', '-Wno-unused-function']
FLAGS = COMPILE + ['-shared']                        # (the one-unit command line: see tools/README.md)
JOBS = max(1, min(8, os.cpu_count() or 1))


def build_id(extra=()):
    '''Digest of what determines the code of the library: sources, header, flags, compiler.'''
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS + list(extra)).encode())      # (round 5's flag string: ids stay comparable)
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


def commands(out, extra):
    '''(compile commands of the families, link command) of one library.'''
    name = os.path.basename(out)
    tag = os.path.splitext(name)[0]
    define = '-DMOD16_BUILD_ID="%s"' % build_id(extra)
    objs, compiles = [], []
    for fam in FAMILIES:
        obj = os.path.join(OBJ, '%s_%s.o' % (tag, os.path.splitext(os.path.basename(fam))[0]))
        objs.append(obj)
        compiles.append([HIPCC] + COMPILE + list(extra) + [define, '-c', fam, '-o', obj])
    link = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-soname,' + name, '-o', out] + objs
    return compiles, link


def build(force=False, verbose=True, experiments=True):
    '''Compiles what is out of date: the entry-point families of capi/ side by side (JOBS at a time),
    then one link per library.'''
    todo = [(OUT, [])] + ([(OUT_EXP, ['-DMOD16_EXPERIMENTS'])] if experiments else [])
    todo = [(out, extra) for out, extra in todo if force or not up_to_date(out, extra)]
    if not todo:
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    plans = [commands(out, extra) for out, extra in todo]
    queue = [cmd for compiles, _ in plans for cmd in compiles]
    running = []
    while queue or running:
        while queue and len(running) < JOBS:
            cmd = queue.pop(0)
            if verbose:
                print(' '.join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, proc = running.pop(0)
        if proc.wait() != 0:
            for _, other in running:
                other.kill()
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    for _, link in plans:
        if verbose:
            print(' '.join(link), flush=True)
        subprocess.check_call(link)
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
