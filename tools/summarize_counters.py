#!/usr/bin/env python3
"""Condenses the passes of tools/prof.sh into profiles/<tag>_kernel_stats.csv and profiles/<tag>_counters.csv
(per kernel: average duration, FETCH_SIZE / WRITE_SIZE and the corrected HBM bytes 2*FETCH + WRITE per launch, the
SQ counters per launch).  Reads rocprofv3's rocpd databases (ROCm 7.2 default output) or its CSV files.
    python tools/summarize_counters.py TAG gpurun_out/prof_TAG"""
import collections
import csv
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace('void ', '').split('(')[0].replace('fep::', '')


def kernel_stats(d):
    """[(name, calls, total_ns, avg_ns, pct, min_ns, max_ns, median_ns, steady_avg_ns)] of one --stats pass.  `steady_avg`
    leaves out a kernel's FIRST launch when it has more than three (code load: round 4's Q2 pass held one 31.5 ms first call
    among 43 of 0.5 ms); it is the figure the counter table divides by."""
    rows = []
    for f in glob.glob(os.path.join(d, '*', '*_kernel_stats.csv')):
        for r in csv.DictReader(open(f)):
            calls, tot, mx = int(r['Calls']), float(r['TotalDurationNs']), float(r['MaxNs'])
            steady = (tot - mx) / (calls - 1) if calls > 3 else tot / max(calls, 1)
            rows.append((short(r['Name']), calls, tot, float(r['AverageNs']), float(r['Percentage']), float(r['MinNs']), mx,
                         float('nan'), steady))
    for f in glob.glob(os.path.join(d, '*', '*.db')):
        c = sqlite3.connect(f)
        tot = c.execute('select sum(duration) from kernels').fetchone()[0] or 1.0
        per = collections.defaultdict(list)
        try:
            cur = c.execute('select name, duration from kernels order by start')
        except sqlite3.OperationalError:                 # (no `start` column in this rocpd version: launch order unknown)
            cur = c.execute('select name, duration from kernels')
        for name, dur in cur:
            per[name].append(float(dur))
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            srt = sorted(v)
            steady = (sum(v) - srt[-1]) / (len(v) - 1) if len(v) > 3 else sum(v) / len(v)     # (the first launch is the longest one)
            rows.append((short(name), len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, srt[0], srt[-1], srt[len(srt) // 2], steady))
    return rows


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '*', '*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(os.path.join(d, '*', '*.db')):
        c = sqlite3.connect(f)
        # one row per (dispatch, counter, dimension instance): sum the instances of a dispatch, average over dispatches
        q = ('select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection '
             'group by kernel_name, counter_name, dispatch_id')
        for k, cn, _, v in c.execute(q):
            acc[short(k)][cn].append(float(v))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    tag, base = sys.argv[1:3]
    latest = '--traffic-latest' in sys.argv          # this run is `bench.py` at its default size: refresh traffic_latest.json
    out = os.path.join(ROOT, 'profiles')
    rows = kernel_stats(os.path.join(base, 'stats'))
    with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'MedianNs', 'AverageNsWithoutFirstLaunch'])
        for r in rows:
            w.writerow([r[0], r[1], f'{r[2]:.0f}', f'{r[3]:.0f}', f'{r[4]:.2f}', f'{r[5]:.0f}', f'{r[6]:.0f}', f'{r[7]:.0f}', f'{r[8]:.0f}'])
    avg = {r[0]: r[8] for r in rows}                      # (steady-state average: without the first launch)
    allc = collections.defaultdict(dict)
    for sub in ('fetch', 'write', 'sq', 'sq2'):
        for k, cs in counters(os.path.join(base, sub)).items():
            allc[k].update(cs)
    names = sorted({c for cs in allc.values() for c in cs})
    with open(os.path.join(out, f'{tag}_counters.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'avg_us', 'hbm_bytes_per_launch=(2*FETCH_SIZE+WRITE_SIZE)KiB*1024', 'GBps_on_those_bytes'] + names)
        for k in sorted(allc, key=lambda k: -avg.get(k, 0.0)):
            if 'kernel' not in k or k.startswith('at::') or 'elementwise' in k or k.startswith('__amd'):
                continue
            cs = allc[k]
            b = (2 * cs.get('FETCH_SIZE', 0.0) + cs.get('WRITE_SIZE', 0.0)) * 1024
            us = avg.get(k, 0.0) / 1e3
            w.writerow([k, f'{us:.2f}', f'{b:.0f}', f'{b / us / 1e3:.0f}' if us else ''] + [f'{cs.get(c, float("nan")):.0f}' for c in names])
            valu = cs.get('SQ_ACTIVE_INST_VALU', float('nan')) / max(cs.get('SQ_BUSY_CYCLES', float('nan')), 1.0)
            print(f'{k[:60]:60s} {us:9.1f} us  {b / 1e6:9.1f} MB  ' + (f'{b / us / 1e3:6.0f} GB/s' if us else '') +
                  f'  VALU-active/busy {valu:.2f}')
    if latest:
        import json
        hb = {k: (2 * cs.get('FETCH_SIZE', 0.0) + cs.get('WRITE_SIZE', 0.0)) * 1024 for k, cs in allc.items()}
        dom = [k for k in hb if k.startswith('p1_point_kernel') or k.startswith('p1_node_lds_kernel')]
        json.dump({'round': tag, 'kernels': {k: hb[k] for k in dom}, 'hbm_bytes_per_launch': sum(hb[k] for k in dom),
                   'kf_only_kernel': {k: hb[k] for k in hb if k.startswith('p1_fused_kernel')},
                   'note': 'full-output step = p1_point_kernel + p1_node_lds_kernel; 2*FETCH_SIZE + WRITE_SIZE (KiB) * 1024, '
                           'separate rocprofv3 --pmc passes of `python3 bench.py --steps 10 --no-cpu-baseline` (tools/prof.sh)',
                   'elements_per_gpu': 1002528, 'element_type': 'P1', 'state': 'bands'},
                  open(os.path.join(out, 'traffic_latest.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
