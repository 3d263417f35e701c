#!/usr/bin/env python3
"""print a compact table (kernel, calls, total ms, avg us, share) from a rocprofv3 kernel_stats.csv; cpe kernels only unless --all"""
import csv
import re
import sys


def short(name):
    m = re.search(r'(k_\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:60]


def main():
    fn = sys.argv[1]
    allk = '--all' in sys.argv
    rows = []
    for r in csv.DictReader(open(fn)):
        if not allk and 'k_' not in r['Name']:
            continue
        rows.append((short(r['Name']), int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
    tot = sum(r[2] for r in rows)
    print(f'{"kernel":40s} {"calls":>7s} {"total ms":>10s} {"avg us":>10s} {"share":>6s}')
    for k, c, ms, us in sorted(rows, key=lambda r: -r[2]):
        print(f'{k:40s} {c:7d} {ms:10.3f} {us:10.1f} {100 * ms / tot:5.1f}%')
    print(f'{"total":40s} {sum(r[1] for r in rows):7d} {tot:10.3f}')


if __name__ == '__main__':
    main()
