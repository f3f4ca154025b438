"""Aggregate rocprofv3 --pmc counter CSVs (one pass per counter) into HBM bytes per launch per kernel.

usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced
reads (16 B/lane loads, incl. buffer_load ... lds; MI355X_MICROARCH.md section HBM), so fetch bytes are doubled;
WRITE_SIZE is taken as is (uncalibrated)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def csrc_stamp():
    """sha256 (first 16 hex digits) over the kernel sources the counters were measured on: bench.py reports the traffic only while
    the sources it runs are the ones stamped here (the GPU box has no .git to ask)"""
    import hashlib
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'open-solution-mapping-challenge_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(here)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(here, f), 'rb').read())
    return h.hexdigest()[:16]


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(anonymous namespace\)::|msc_conv::', '', r['Kernel_Name'])
            name = re.sub(r'^void ', '', name).split('(')[0]
            acc[name][r['Counter_Name']] += float(r['Counter_Value'])
            calls[name][r['Counter_Name']] += 1
    rows = []
    for name, cs in acc.items():
        n_f, n_w = calls[name].get('FETCH_SIZE', 0), calls[name].get('WRITE_SIZE', 0)
        fetch = cs.get('FETCH_SIZE', 0.0) * 1024 * 2 / max(n_f, 1)
        write = cs.get('WRITE_SIZE', 0.0) * 1024 / max(n_w, 1)
        rows.append({'kernel': name, 'launches': max(n_f, n_w), 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                     'hbm_bytes_per_launch': fetch + write})
    rows.sort(key=lambda r: -r['hbm_bytes_per_launch'] * r['launches'])
    FAMILY = ('conv_igemm_dma_kernel', 'conv3x3_halo_dma_kernel', 'conv3x3_c32_halo_kernel', 'deconv4_c128_c32_halo_kernel', 'conv1x1_stream_kernel',
              'stem7_halo_kernel', 'splitk_finish_kernel')      # = msc_conv_igemm
    fam = [r for r in rows if r['kernel'].startswith(FAMILY)]
    tot_l = sum(r['launches'] for r in fam)
    summary = {'family': 'msc_conv_igemm (conv_igemm_dma_kernel + halo-tile kernels)', 'launches': tot_l, 'csrc_sha16': csrc_stamp(),
               'hbm_bytes_per_launch': sum(r['hbm_bytes_per_launch'] * r['launches'] for r in fam) / max(tot_l, 1),
               'note': 'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, averaged over all launches of the family',
               'kernels': rows[:40]}
    print(json.dumps({k: v for k, v in summary.items() if k != 'kernels'}, indent=1))
    for r in rows[:25]:
        print('%-90s x%5d  fetch %10.2f MB  write %10.2f MB' % (r['kernel'][:90], r['launches'], r['fetch_bytes_per_launch'] / 1e6, r['write_bytes_per_launch'] / 1e6))
    if len(sys.argv) > 2:
        json.dump(summary, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
