#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/...) into the small tracked files under profiles/.

    python tools/summarize_profile.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [elements per GPU of the profiled run]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table, fep kernels + totals),
profiles/<tag>_hbm_traffic.csv (per kernel FETCH_SIZE / WRITE_SIZE averages and corrected bytes) and
profiles/traffic_latest.json (bytes per launch of the dominant return-map + assembly kernels; read by
bench.py for roofline.traffic).

HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes x1024): on gfx950 FETCH_SIZE tallies 128-byte
requests as 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.  The factor was checked
here on p1_point_kernel, whose read set is known (elements 12 B + geometry record 64 B + materials
32 B + plastic strain 32 B + displacements ~8 B per element = 148 MB at 1 002 528 elements):
2 * FETCH_SIZE = 154 MB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace('void ', '')
    return n.split('(')[0]


def pmc(d, counter):
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            acc[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    tag, stats_dir, fdir, wdir = sys.argv[1:5]
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    f = glob.glob(os.path.join(stats_dir, '*', '*_kernel_stats.csv'))[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        for r in rows:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'],
                        r['MaxNs'], r['StdDev']])
    fetch, write = pmc(fdir, 'FETCH_SIZE'), pmc(wdir, 'WRITE_SIZE')
    traffic = {}
    with open(os.path.join(out, f'{tag}_hbm_traffic.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'FETCH_SIZE_KiB_avg', 'WRITE_SIZE_KiB_avg', 'hbm_bytes_per_launch(2*FETCH+WRITE)'])
        for k in sorted(set(fetch) | set(write)):
            if 'fep::' not in k:
                continue
            b = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024
            traffic[k] = b
            w.writerow([k, f'{fetch.get(k, 0.0):.1f}', f'{write.get(k, 0.0):.1f}', f'{b:.0f}'])
    dom = [k for k in traffic if any(s in k for s in ('p1_point_kernel', 'p1_node', 'p1_fused', 'element_kernel'))]
    if os.environ.get('FEP_SUMMARY_ONLY'):       # other workloads than bench.py's: tables only, traffic_latest.json untouched
        print(json.dumps({k: round(v / 1e6, 1) for k, v in traffic.items()}))
        return
    json.dump({'round': tag, 'kernels': {k: traffic[k] for k in dom},
               'hbm_bytes_per_launch': sum(traffic[k] for k in dom),
               'note': 'sum over the return-map + assembly kernels of one step; 2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes',
               'elements_per_gpu': int(sys.argv[5]) if len(sys.argv) > 5 else 1002528},
              open(os.path.join(out, 'traffic_latest.json'), 'w'), indent=1)
    for r in rows[:6]:
        print(short(r['Name'])[:60], r['Calls'], r['AverageNs'], r['Percentage'])
    print(json.dumps({k: round(v / 1e6, 1) for k, v in traffic.items()}))


if __name__ == '__main__':
    main()
