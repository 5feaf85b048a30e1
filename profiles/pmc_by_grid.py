#!/usr/bin/env python
"""Per-(kernel instantiation, grid size) totals of one rocprofv3 --pmc counter: which launch shapes carry the bytes.
usage: pmc_by_grid.py DIR [top]   (DIR holds *counter_collection.csv)"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'void ([a-z_0-9]+)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:48]


def main(d, top=40):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    cname = None
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r['Kernel_Name']), int(r['Grid_Size']) // max(1, int(r.get('Workgroup_Size', 256) or 256)))
            tot[k] += float(r['Counter_Value'])
            cnt[k] += 1
            cname = r['Counter_Name']
    all_ = sum(tot.values())
    print('# %s: total %.4g over %d dispatches; rows: kernel, workgroups, dispatches, total, per dispatch, share' % (cname, all_, sum(cnt.values())))
    for k in sorted(tot, key=lambda k: -tot[k])[:top]:
        print('%-52s %7d %5d %12.5g %12.5g %6.2f%%' % (k[0][:52], k[1], cnt[k], tot[k], tot[k] / cnt[k], 100 * tot[k] / all_))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
