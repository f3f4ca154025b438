"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (one directory per pass, *_counter_collection.csv).
usage: python tools/pmc_sq_summary.py <dir> [<dir> ...]   -> table: kernel, launches, grid, one column per counter"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
    grid = {}
    for d in sys.argv[1:]:
        files = sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True), key=os.path.getmtime)
        if not files:
            continue
        for r in csv.DictReader(open(files[-1])):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            name = re.sub(r'^void ', '', name)
            name = re.sub(r'\(.*$', '', name)[:70]
            key = (name, r['Grid_Size'])
            per[key][r['Counter_Name']].append(float(r['Counter_Value']))
            grid[key] = (int(r['Grid_Size']), int(r['Workgroup_Size']))
    counters = sorted({c for v in per.values() for c in v})
    print('%-72s %6s %8s ' % ('kernel', 'n', 'blocks') + ' '.join('%16s' % c[-16:] for c in counters))
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0])))
    for key, cs in rows[:60]:
        n = max(len(v) for v in cs.values())
        g, wg = grid[key]
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        line = '%-72s %6d %8d ' % (key[0], n, g // max(wg, 1)) + ' '.join('%16.0f' % avg[c] if c in cs else '%16s' % '-' for c in counters)
        wc = avg.get('SQ_WAVE_CYCLES', 0)
        if wc:      # shares of the wave cycles: parked (s_waitcnt / barrier), issue-stalled, issuing; MFMA-busy cycles per wave cycle (x waves per SIMD = pipe utilisation)
            line += '   parked %4.1f%%  stalled %4.1f%%  issuing %4.1f%%  mfma/wave-cycle %5.3f' % (
                100 * avg.get('SQ_WAIT_ANY', 0) / wc, 100 * avg.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * avg.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * wc))
        print(line)


if __name__ == '__main__':
    main()
