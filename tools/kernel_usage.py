"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: python tools/kernel_usage.py [file.hip ...] [--grep enc_]   (default: every source of libpiml_hip.so)"""
import os
import re
import subprocess
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from piml_amd import build as B  # noqa: E402


def usage(srcs):
    cmd = ['/opt/rocm/bin/hipcc'] + [f for f in B.FLAGS if f not in ('-shared',)] + \
        os.environ.get('PIML_HIPCC_EXTRA', '').split() + ['-Rpass-analysis=kernel-resource-usage', '-c', '-o', '/dev/null']
    rows = []
    for src in srcs:
        out = subprocess.run(cmd + [src], capture_output=True, text=True).stderr
        cur = None
        for line in out.splitlines():
            m = re.search(r'Function Name: (\S+)', line)
            if m:
                cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
                rows.append(cur)
                continue
            for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                             ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('spill', r'VGPRs Spill: (\d+)'),
                             ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    return rows


if __name__ == '__main__':
    args = sys.argv[1:]
    pat = None
    if '--grep' in args:
        i = args.index('--grep')
        pat = args[i + 1]
        del args[i:i + 2]
    rows = usage(args or B.sources())
    print(f'{"kernel":90s} vgpr agpr spill scratch occ')
    for r in rows:
        if pat and pat not in r['name']:
            continue
        name = re.sub(r'\(.*', '', r['name'])[:90]
        print(f'{name:90s} {r.get("vgpr", -1):4d} {r.get("agpr", -1):4d} {r.get("spill", -1):5d} {r.get("scratch", -1):7d} {r.get("occ", -1):3d}')
