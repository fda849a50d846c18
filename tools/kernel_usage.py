"""Per-kernel register / scratch / occupancy table of libpiml_hip.so from `hipcc -Rpass-analysis=kernel-resource-usage`
(the per-file flags of piml_amd/build.py apply).
usage: python -m piml_amd.build --usage 2> usage.log; python tools/kernel_usage.py usage.log [--md] [--spills]"""
import re
import subprocess
import sys


def parse(path):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'mangled': m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('spill', r'VGPRs Spill: (\d+)')):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = subprocess.run(['c++filt'] + [r['mangled'] for r in rows], capture_output=True, text=True).stdout.splitlines()
    seen, out = set(), []
    for r, n in zip(rows, names):
        r['name'] = re.sub(r'\(.*', '', n).replace('void ', '').replace('piml::', '')
        if r['name'] not in seen:
            seen.add(r['name'])
            out.append(r)
    return out


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rows = parse(args[0])
    if '--spills' in sys.argv:
        rows = [r for r in rows if r.get('spill', 0) > 0]
    if '--md' in sys.argv:
        print('| kernel | VGPRs | AGPRs | VGPRs spilled | scratch B/lane | waves/SIMD |\n|---|---|---|---|---|---|')
        for r in rows:
            print(f"| `{r['name']}` | {r.get('vgpr')} | {r.get('agpr')} | {r.get('spill')} | {r.get('scratch')} | {r.get('occ')} |")
    else:
        for r in rows:
            print(f"{r['name'][:100]:100s} vgpr {r.get('vgpr'):4d} agpr {r.get('agpr'):4d} spill {r.get('spill'):4d} scratch {r.get('scratch'):5d} occ {r.get('occ')}")
