#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel family: sum of each counter over all dispatches."""
import collections
import csv
import glob
import re
import sys


def fam(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'void ([a-z_0-9]+)', name)
    return m.group(1) if m else name[:40]


def main(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = fam(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (k, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    for k in sorted(agg, key=lambda k: -cnt[k]):
        print('%-28s dispatches %6d  ' % (k, cnt[k]) + '  '.join('%s=%.6g' % kv for kv in sorted(agg[k].items())))


if __name__ == '__main__':
    main(sys.argv[1])
