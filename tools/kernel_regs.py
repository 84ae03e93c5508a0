"""VGPRs / scratch bytes / LDS of every kernel in a hipcc -S listing.   python tools/kernel_regs.py file.s [filter]"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    v = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
    sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)
    rows.append((name, v, sc))
dn = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().splitlines()
for (name, v, sc), d in zip(rows, dn):
    if flt in d:
        print("%4s vgpr %5s B scratch  %s" % (v, sc, d[:150]))
