"""Scan a device assembly listing (hipcc -S --cuda-device-only) for kernels whose global loads are waited on one at a time
(many `s_waitcnt vmcnt(0)` per load: a load -> use -> store chain the compiler could not reorder, e.g. for aliasing).
usage: python scripts/isa_wait_scan.py x.s"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
parts = re.split(r'\n(_Z\w+): +; @\w+\n', t)
rows = []
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('s_endpgm')[0]
    ld = len(re.findall(r'global_load', body)); st = len(re.findall(r'global_store', body))
    w0 = len(re.findall(r's_waitcnt vmcnt\(0\)', body)); w = len(re.findall(r's_waitcnt vmcnt', body))
    if ld + st < 4:
        continue
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0][:64]
    rows.append((w0 / (ld + 1.0), dn, ld, st, w, w0, len(body.splitlines())))
print("%-5s %-64s %6s %6s %6s %6s %7s" % ("w0/ld", "kernel", "loads", "stores", "waits", "wait0", "lines"))
for r in sorted(rows, reverse=True):
    print("%.2f  %-64s %6d %6d %6d %6d %7d" % r)
