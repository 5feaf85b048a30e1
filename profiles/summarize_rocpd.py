#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run stored as rocpd sqlite."""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
    scols = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
    name_col = 'kernel_name' if 'kernel_name' in scols else ('display_name' if 'display_name' in scols else scols[-1])
    q = 'select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc' % (name_col, kd, ks, name_col)
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    print('%-100s %8s %12s %10s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'share'))
    for name, n, t, mn, mx in rows[:top]:
        short = re.sub(r'\(anonymous namespace\)::', '', name)
        short = re.sub(r'\(ConvParams.*', '', short)
        print('%-100s %8d %12.3f %10.2f %6.1f%%' % (short[:100], n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
    print('TOTAL kernel time %.3f ms over %d dispatches' % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
