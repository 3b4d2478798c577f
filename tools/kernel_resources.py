#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of libfep_hip.so as hipcc reports it
(-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).  `python tools/kernel_resources.py [filter]`."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fem-elastoplasticity_amd', 'csrc')


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    rows = []
    for src in ('fep_api.hip', 'fep_solver.hip'):
        out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                              '-Rpass-analysis=kernel-resource-usage', '-o', '/dev/null', src], cwd=CSRC,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        cur = None
        for line in out.splitlines():
            m = re.search(r'remark: [^ ]* +(Function Name|Name): (\S+)', line)
            if m:
                name = subprocess.run(['c++filt', m.group(2)], stdout=subprocess.PIPE, text=True).stdout.strip()
                cur = {'name': re.sub(r'\(.*', '', name).replace('fep::', '')}
                rows.append(cur)
                continue
            m = re.search(r'remark: [^ ]* +(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
            if m and cur is not None:
                cur[m.group(1).split(' ')[0]] = int(m.group(2))
    print(f'{"kernel":70s} {"VGPR":>5s} {"AGPR":>5s} {"SGPR":>5s} {"scratch":>7s} {"occ":>4s} {"LDS":>7s}')
    for r in rows:
        if flt in r['name']:
            print(f'{r["name"][:70]:70s} {r.get("VGPRs", 0):5d} {r.get("AGPRs", 0):5d} {r.get("TotalSGPRs", 0):5d} '
                  f'{r.get("ScratchSize", 0):7d} {r.get("Occupancy", 0):4d} {r.get("LDS", 0):7d}')


if __name__ == '__main__':
    main()
