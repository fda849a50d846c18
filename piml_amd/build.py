"""Build recipe for libpiml_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the
repository snapshot).  `python -m piml_amd.build` or `__graft_entry__.build()`.

Every csrc/*.hip is compiled to an object of its own under piml_amd/_obj/ (git-ignored; re-made only when the source, a
header or the flags changed) and the objects are linked into the library: an edit of one kernel file costs seconds, not
a minute.  `variant(name, {file: [-D flags]})` links an EXPERIMENTAL library beside the shipped one from the same
objects with some files re-compiled under extra flags (tools/: A/B timings, "results wrong on purpose" builds); it is
selected at run time with PIML_LIB=<path> (piml_amd/_lib.py) and never overwrites libpiml_hip.so."""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libpiml_hip.so')

# -ffp-contract=off: selection predicates must evaluate exactly the float32 operations the
# reference's CPU kernels do; the only fused multiply-adds are the explicit __fmaf_rn calls.
CFLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off',
          '-fhip-fp32-correctly-rounded-divide-sqrt',
          '-fvisibility=hidden', '-Wall', '-Wno-unused-function']
FLAGS = CFLAGS + ['-shared']      # (the one-command form; kept for readers of older notes)
# per-file extras.  encoder_bwd3.hip: its vector work sits between matrix instructions issued by ONE wave per SIMD, where a
# packed-f32 instruction (v_pk_fma_f32 / v_pk_add_f32, what the SLP vectoriser makes of two neighbouring scalar operations)
# costs ~16 issue cycles against 2 x 4 (tools/probe_mfma_slots.hip).
FILE_FLAGS = {'encoder_bwd3.hip': ['-fno-slp-vectorize'], 'encoder_bwd4.hip': ['-fno-slp-vectorize'],
              'encoder_bwd5.hip': ['-fno-slp-vectorize']}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.hpp')) + glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h')))


def _hipcc():
    return os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _stamp(src, flags):
    """identity of an object: its source, every header (a header edit re-makes every object) and the flags"""
    h = hashlib.sha256()
    for f in [src] + _headers():
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def _compile(src, flags, tag='', verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    base = os.path.join(OBJ, os.path.basename(src)[:-4] + tag)
    obj, stamp_file = base + '.o', base + '.stamp'
    stamp = _stamp(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [_hipcc()] + flags + ['-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, 'w') as fh:
        fh.write(stamp)
    return obj


def _link(objs, out, verbose=False):
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def build(force=False, verbose=False, extra=()):
    if not force and not extra and not needs_build():
        return LIB
    flags = CFLAGS + list(extra) + os.environ.get('PIML_HIPCC_EXTRA', '').split()
    with ThreadPoolExecutor(max_workers=int(os.environ.get('PIML_BUILD_JOBS', '4'))) as ex:
        objs = list(ex.map(lambda s: _compile(s, flags + FILE_FLAGS.get(os.path.basename(s), []), verbose=verbose, force=bool(extra)), sources()))
    return _link(objs, LIB, verbose)


def variant(name, defines, verbose=False):
    """libpiml_hip_<name>.so: the shipped objects, with the files named in `defines` ({'encoder_bwd3.hip': ['-DX=1']})
    re-compiled under the extra flags.  Returns the path (pass it as PIML_LIB)."""
    flags = CFLAGS + os.environ.get('PIML_HIPCC_EXTRA', '').split()
    objs = []
    for s in sources():
        extra = defines.get(os.path.basename(s))
        ff = flags + FILE_FLAGS.get(os.path.basename(s), [])
        objs.append(_compile(s, ff + list(extra), tag='.' + name, verbose=verbose) if extra else _compile(s, ff, verbose=verbose))
    return _link(objs, os.path.join(HERE, f'libpiml_hip_{name}.so'), verbose)


if __name__ == '__main__':
    if '--variant' in sys.argv:          # python -m piml_amd.build --variant NAME file.hip:-DA=1,-DB file2.hip:-DC
        i = sys.argv.index('--variant')
        spec = {a.split(':', 1)[0]: a.split(':', 1)[1].split(',') for a in sys.argv[i + 2:]}
        print(variant(sys.argv[i + 1], spec, verbose=True))
    else:
        build(force='--force' in sys.argv, verbose=True,
              extra=['-Rpass-analysis=kernel-resource-usage'] if '--usage' in sys.argv else ())
        print(LIB)
