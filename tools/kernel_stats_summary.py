"""Summarise `rocprofv3 --kernel-trace --stats --output-format csv` output: the per-kernel table of *_kernel_stats.csv
(largest process of the run) plus aggregates per kernel family, which is what bench.py's `roofline.avg_launch_us` has to
agree with.     usage: python tools/kernel_stats_summary.py <rocprof output dir> [top_n]"""
import csv
import glob
import os
import re
import sys

FAMILIES = [('conv_igemm / halo (conv, dgrad, deconv)', r'conv_igemm|halo_kernel|conv3x3_|deconv4_'),
            ('conv_wgrad', r'conv_wgrad|wgrad_finish'),
            ('BatchNorm elementwise (bn_apply, bn_bwd_apply, stem bn + pool)', r'bn_apply_kernel|bn_bwd_apply_kernel|bn_apply_pool_kernel|bn_pool_bwd_kernel'),
            ('column reductions (BN backward sums, bias grads)', r'colreduce'),
            ('BatchNorm finalize kernels', r'bn_finalize|bn_bwd_finalize|bias_finalize'),
            ('post-processing (resize, threshold, ccl, dilate, score)', r'resize_|threshold_|ccl_|rect_filter|score_|dropped_|crop_|argmax_'),
            ('annotation encoding', r'seg_|transpose_cm|rocprim|DeviceRadix|device_scan|lookback'),
            ('optimizer + weight packing', r'adam|pack_multi|stem_pack|grad_check'),
            ('loss + final 1x1', r'loss_|final_')]


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    files = glob.glob(os.path.join(root, '**', '*_kernel_stats.csv'), recursive=True)
    if not files:
        print('no *_kernel_stats.csv under', root)
        return 1
    path = max(files, key=os.path.getsize)
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r['Name'], int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']), float(r['MinNs']), float(r['MaxNs'])))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print('source: %s' % os.path.relpath(path, root))
    print('%-10s %7s %8s %11s %11s %11s  %s' % ('total_ms', 'pct', 'calls', 'avg_us', 'min_us', 'max_us', 'kernel'))
    for name, calls, total, avg, mn, mx in rows[:top]:
        name = re.sub(r'\(anonymous namespace\)::|msc_conv::', '', name)
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name)
        print('%-10.3f %6.2f%% %8d %11.1f %11.1f %11.1f  %s' % (total / 1e6, 100.0 * total / tot, calls, avg / 1e3, mn / 1e3, mx / 1e3, name[:140]))
    print('total kernel time: %.3f ms over %d launches' % (tot / 1e6, sum(r[1] for r in rows)))
    print()
    print('family aggregates (all launches of the run, warm-up and breakdown passes included):')
    print('%-62s %9s %9s %10s' % ('family', 'launches', 'total_ms', 'avg_us'))
    for label, pat in FAMILIES:
        sel = [r for r in rows if re.search(pat, r[0])]
        if sel:
            calls, total = sum(r[1] for r in sel), sum(r[2] for r in sel)
            print('%-62s %9d %9.3f %10.2f' % (label, calls, total / 1e6, total / 1e3 / calls))
    return 0


if __name__ == '__main__':
    sys.exit(main())
