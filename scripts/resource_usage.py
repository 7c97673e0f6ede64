"""Tabulate per-kernel register / scratch / occupancy figures from `hipcc -Rpass-analysis=kernel-resource-usage` output.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> ru.txt ; python scripts/resource_usage.py ru.txt"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', t)[1:]
keys = [('VGPRs', 'vgpr'), ('AGPRs', 'agpr'), (r'ScratchSize \[bytes/lane\]', 'scratch'), (r'Occupancy \[waves/SIMD\]', 'occ'),
        (r'LDS Size \[bytes/block\]', 'lds')]
for b in blocks:
    name = b.split(' [')[0]
    try:
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    vals = []
    for k, lab in keys:
        m = re.search(k + r': (\d+)', b)
        vals.append('%s=%-5s' % (lab, m.group(1) if m else '?'))
    print(' '.join(vals), name.split('(')[0][:80])
