import sys, collections
def load(path):
    agg = collections.OrderedDict()
    for l in open(path):
        r = l.strip().split(',')
        if len(r) < 10: continue
        a = agg.setdefault(tuple(r[:9]), [0, 0.0]); a[0] += 1; a[1] += float(r[9])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in b.values())
print('shape (Cout,nh,K,M,kh,s,os,relu,acc) | n | A avg_us TF | B avg_us TF | B/A')
for k, (n, ms) in sorted(a.items(), key=lambda kv: -kv[1][1]):
    if k not in b: continue
    fl = 2.0 * int(k[2]) * int(k[3]) * int(k[0]) * int(k[1])
    ua, ub = ms / n * 1e3, b[k][1] / b[k][0] * 1e3
    print(','.join(k), '|', n, '| %.1f %.1f | %.1f %.1f | %.3f' % (ua, fl / ua / 1e6, ub, fl / ub / 1e6, ub / ua))
print('total per step A %.3f B %.3f' % (ta, tb))
