"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel dispatch count and mean/total counter value.
    python tools/pmc_summary.py <counter_collection.csv> <COUNTER>"""
import csv
import collections
import re
import sys


def main():
    path, ctr = sys.argv[1], sys.argv[2]
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != ctr:
                continue
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:100]
            a = agg.setdefault(k, [0, 0.0, 0.0])
            v = float(r['Counter_Value'])
            a[0] += 1
            a[1] += v
            a[2] = max(a[2], v)
    print('# %s from %s (raw counter units as rocprofv3 reports them; see profiles/README.md for the gfx950 correction)' % (ctr, path.split('/')[-1]))
    print('%-100s %8s %16s %16s %16s' % ('kernel', 'calls', 'mean', 'max', 'total'))
    for k, (n, s, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-100s %8d %16.1f %16.1f %16.1f' % (k, n, s / n, mx, s))


if __name__ == '__main__':
    main()
