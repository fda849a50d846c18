"""top kernels of a rocprofv3 --stats output directory: python tools/top_kernels.py <dir> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    us = float(r['TotalDurationNs']) / 1e3 / max(int(r['Calls']), 1)
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {us:8.1f} us  {float(r['Percentage']):5.1f} %")
