#!/usr/bin/env python3
"""Tabulate `hipcc -Rpass-analysis=kernel-resource-usage` output: VGPRs / scratch / waves per SIMD / LDS per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt; kernel_resources.py res.txt [name-filter]"""
import re, subprocess, sys
cur, rows = None, {}
for l in open(sys.argv[1]):
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    for k, nm in ((r'VGPRs', 'vgpr'), (r'ScratchSize \[bytes/lane\]', 'scratch'), (r'Occupancy \[waves/SIMD\]', 'occ'),
                  (r'LDS Size \[bytes/block\]', 'lds')):
        m = re.search(r'remark:\s+' + k + r': (\d+)', l)
        if m and cur:
            rows[cur][nm] = int(m.group(1))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for k, v in rows.items():
    d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if flt in d:
        print(d.replace('zk::', '').replace('void ', '')[:90].ljust(90), v)
