#!/usr/bin/env python3
"""Resource table from a saved -Rpass-analysis=kernel-resource-usage log: python tools/kres.py LOG [filter]"""
import re, subprocess, sys
rows = []; cur = None
for line in open(sys.argv[1]):
    m = re.search(r'remark: [^ ]* +(Function Name|Name): (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', name).replace('fep::', '').replace('void ', '')}; rows.append(cur); continue
    m = re.search(r'remark: [^ ]* +(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur is not None and m.group(1).split(' ')[0] not in cur: cur[m.group(1).split(' ')[0]] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in rows:
    if flt in r['name']:
        print(f"{r['name'][:64]:64s} V{r.get('VGPRs',0):4d} A{r.get('AGPRs',0):4d} S{r.get('TotalSGPRs',0):4d} scr{r.get('ScratchSize',0):4d} occ{r.get('Occupancy',0):3d} LDS{r.get('LDS',0):7d}")
