"""Summarise every counter of a rocprofv3 --pmc counter_collection.csv per kernel: mean value per dispatch.
    python tools/pmc_multi.py <counter_collection.csv> [kernel substring ...]"""
import collections
import csv
import re
import sys


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    agg = collections.OrderedDict()
    names = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
            if subs and not any(s in k for s in subs):
                continue
            k = '%s grid=%s' % (k, r.get('Grid_Size', '?'))
            c = r['Counter_Name']
            if c not in names:
                names.append(c)
            a = agg.setdefault(k, {}).setdefault(c, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    for k, d in agg.items():
        print(k)
        for c in names:
            if c in d:
                print('    %-44s calls %4d  mean %18.1f' % (c, d[c][0], d[c][1] / d[c][0]))


if __name__ == '__main__':
    main()
