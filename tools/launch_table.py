"""per-launch dump of bench.py --dump-launches -> conv launches grouped by shape: count, us, TFLOP/s, the time at 2.5 PFLOP/s / at 6.3 TB/s of
algorithmic bytes, and the time lost against max(those two) -- sorted by lost time (where a faster kernel would pay most)."""
import json, sys, collections
rows = json.load(open(sys.argv[1]))
es = 2
g = collections.OrderedDict()
for r in rows:
    if r['k'] == 'conv':
        key = ('conv', r['mode'], r['flip'], r['N'], r['Hi'], r['Wi'], r['Cin'], r['Ho'], r['Wo'], r['Cout'], r['KH'], r['KW'], r['stride'], r['stats'], r['res'])
        flops = 2.0 * r['N'] * r['Ho'] * r['Wo'] * r['Cout'] * r['Cin'] * r['KH'] * r['KW'] / (4.0 if r['mode'] == 1 else 1.0)
        byts = es * (r['N'] * r['Hi'] * r['Wi'] * r['Cin'] + r['N'] * r['Ho'] * r['Wo'] * r['Cout'] * (2 if r['res'] else 1) + r['Cout'] * r['Cin'] * r['KH'] * r['KW'])
    elif r['k'] == 'wgrad':
        key = ('wgrad', r['N'], r['Hp'], r['Wp'], r['A'], r['Hq'], r['Wq'], r['B'], r['KH'], r['KW'], r['stride'])
        flops = 2.0 * r['N'] * r['Hp'] * r['Wp'] * r['A'] * r['B'] * r['KH'] * r['KW']
        byts = es * (r['N'] * r['Hp'] * r['Wp'] * r['A'] + r['N'] * r['Hq'] * r['Wq'] * r['B']) + 4 * r['A'] * r['B'] * r['KH'] * r['KW']
    else:
        key = ('bneck', r['N'], r['H'], r['W'], r['Cmid'])
        flops = 2.0 * r['N'] * r['H'] * r['W'] * 17 * r['Cmid'] ** 2
        byts = es * (2 * r['N'] * r['H'] * r['W'] * 4 * r['Cmid'] + 17 * r['Cmid'] ** 2)
    e = g.setdefault(key, {'n': 0, 'us': 0.0, 'flops': flops, 'bytes': byts})
    e['n'] += 1
    e['us'] += r['us']
tot = sum(e['us'] for e in g.values())
print('%d launches, %d shapes, %.3f ms' % (sum(e['n'] for e in g.values()), len(g), tot / 1e3))
out = []
for key, e in g.items():
    us = e['us'] / e['n']
    t_mfma, t_hbm = e['flops'] / 2.5e15 * 1e6, e['bytes'] / 6.3e12 * 1e6
    floor = max(t_mfma, t_hbm) + 2.0                       # + the dependent-launch floor
    out.append((e['n'] * (us - floor), key, e['n'], us, e['flops'] / us / 1e6, t_mfma, t_hbm))
print('lost_us   n   us/launch  TFLOP/s  t_mfma  t_hbm   shape')
for lost, key, n, us, tf, tm, th in sorted(out, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%7.0f %3d %10.1f %8.0f %7.1f %6.1f   %s' % (lost, n, us, tf, tm, th, key))
