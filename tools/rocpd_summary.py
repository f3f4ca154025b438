"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / average, like --stats.
usage: python tools/rocpd_summary.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = list(cur.execute('select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                            'from kernels group by %s order by 3 desc' % (name_col, name_col)))
    tot = sum(r[2] for r in rows)
    print('%-10s %7s %8s %11s %11s %11s  %s' % ('total_ms', 'pct', 'calls', 'avg_us', 'min_us', 'max_us', 'kernel'))
    for name, calls, total, avg, mn, mx in rows[:top]:
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'^void ', '', name)
        print('%-10.3f %6.2f%% %8d %11.1f %11.1f %11.1f  %s' % (total / 1e6, 100.0 * total / tot, calls, avg / 1e3, mn / 1e3, mx / 1e3, name[:150]))
    print('total kernel time: %.3f ms over %d kernels' % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == '__main__':
    main()
