"""Build recipe for libpiml_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the
repository snapshot).  `python -m piml_amd.build` or `__graft_entry__.build()`."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpiml_hip.so')

# -ffp-contract=off: selection predicates must evaluate exactly the float32 operations the
# reference's CPU kernels do; the only fused multiply-adds are the explicit __fmaf_rn calls.
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-std=c++17', '-ffp-contract=off',
         '-fhip-fp32-correctly-rounded-divide-sqrt',
         '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.hpp')) + \
        glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + list(extra) + os.environ.get('PIML_HIPCC_EXTRA', '').split() + ['-o', LIB] + sources()
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True,
          extra=['-Rpass-analysis=kernel-resource-usage'] if '--usage' in sys.argv else ())
    print(LIB)
