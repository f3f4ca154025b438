"""rocprofv3 --kernel-trace csv -> per (kernel, grid, workgroup) count / mean / min duration, sorted by total time."""
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get('Kernel_Name') or r.get('kernel_name')
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        g = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Z', ''), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
        name = name.replace('(anonymous namespace)::', '').replace('void ', '')
        rows[(name.split('(')[0][:100], g)].append(d)
tot = sum(sum(v) for v in rows.values())
print('total kernel time %.3f ms' % (tot / 1e3))
for (name, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print('%9.1f us tot %6d calls  mean %8.2f  med %8.2f  min %8.2f  grid %-10s z %-2s wg %-4s %s' % (sum(v), len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], g[0], g[1], g[2], name))
