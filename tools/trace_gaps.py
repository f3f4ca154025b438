"""rocprofv3 --kernel-trace csv -> where the GPU idles between kernels: every gap above a threshold with the kernels on both sides,
the gap histogram, and the busy / wall ratio of the last steps (all queues merged: a gap is time in which NO kernel ran)."""
import csv, glob, sys
thr_us = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = (r.get('Kernel_Name') or r.get('kernel_name')).replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', '?')))
rows.sort()
rows = rows[len(rows) // 2:]                      # the second half of the run: steady-state replays
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy_end, busy, gaps = rows[0][0], 0, []
for s, e, name, q in rows:
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, prev, name, q))
        busy += 0
    busy += max(0, e - max(s, busy_end))
    if e > busy_end:
        busy_end, prev = e, name
print('kernels %d  wall %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)' % (len(rows), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, 100.0 * (t1 - t0 - busy) / (t1 - t0)))
big = [g for g in gaps if g[0] >= thr_us]
print('gaps >= %.1f us: %d, total %.3f ms; gaps below: %d, total %.3f ms' % (thr_us, len(big), sum(g[0] for g in big) / 1e3, len(gaps) - len(big), sum(g[0] for g in gaps if g[0] < thr_us) / 1e3))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b, q in big:
    agg[(a, b)][0] += 1
    agg[(a, b)][1] += g
for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%4d x %8.1f us avg  after %-70s before %s' % (n, t / n, a, b))
