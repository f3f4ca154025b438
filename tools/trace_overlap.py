"""rocprofv3 --kernel-trace csv of the train step -> how much of the weight-gradient kernels' time overlaps other kernels, per step.
usage: trace_overlap.py <dir> [n_last_steps]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = (r.get('Kernel_Name') or r.get('kernel_name')).replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
is_w = lambda n: 'wgrad' in n
# step boundaries: the adam_pack kernel closes a step
steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'adam_pack' in r[2]:
        steps.append(cur); cur = []
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for st in steps[-nlast:]:
    t0, t1 = st[0][0], max(r[1] for r in st)
    w = [r for r in st if is_w(r[2])]
    o = [r for r in st if not is_w(r[2])]
    wsum = sum(r[1] - r[0] for r in w) / 1e3
    osum = sum(r[1] - r[0] for r in o) / 1e3
    # overlap: time during which a wgrad kernel and a non-wgrad kernel are both running
    ev = []
    for r in w: ev += [(r[0], 0, 1), (r[1], 0, -1)]
    for r in o: ev += [(r[0], 1, 1), (r[1], 1, -1)]
    ev.sort()
    c = [0, 0]; last = ev[0][0]; both = 0; anyk = 0
    for t, k, d in ev:
        if c[0] > 0 and c[1] > 0: both += t - last
        if c[0] > 0 or c[1] > 0: anyk += t - last
        c[k] += d; last = t
    queues = sorted(set(r[3] for r in st))
    print('step %.3f ms  busy %.3f  wgrad kernels %.3f ms (%d)  others %.3f ms (%d)  both running %.3f ms  queues %s' %
          ((t1 - t0) / 1e6, anyk / 1e6, wsum / 1e3, len(w), osum / 1e3, len(o), both / 1e6, queues))
st = steps[-1]
print('--- wgrad launches of the last step (start offset ms, dur us, queue, name) and the kernels running beside them')
t0 = st[0][0]
for r in st:
    if is_w(r[2]):
        beside = [x for x in st if not is_w(x[2]) and x[0] < r[1] and x[1] > r[0]]
        bs = sum(min(x[1], r[1]) - max(x[0], r[0]) for x in beside) / 1e3
        print('%8.3f %9.1f q%s %s | beside: %d kernels, %.1f us' % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e3, r[3], r[2][:70], len(beside), bs))
