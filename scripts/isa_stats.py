#!/usr/bin/env python3
"""scripts/isa_stats.py a.s [b.s] -- per kernel of a `hipcc --cuda-device-only -S` listing: VGPRs, scratch bytes, LDS bytes, vector /
scalar instruction counts (static); with two listings, side by side (A/B of a source change without a GPU)."""
import re
import subprocess
import sys


def parse(p):
    txt = open(p).read()
    out = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(r'\.amdhsa_' + k + r' (\d+)', body).group(1))
        out[m.group(1)] = [g('next_free_vgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), 0, 0]
    for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
        if m.group(1) in out:
            out[m.group(1)][3] = len(re.findall(r'^\s+v_', m.group(2), re.M))
            out[m.group(1)][4] = len(re.findall(r'^\s+s_', m.group(2), re.M))
    return out


def demangle(names):
    try:
        r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
        return {n: d.split('(')[0].replace('void ', '') for n, d in zip(names, r)}
    except OSError:
        return {n: n for n in names}


a = parse(sys.argv[1])
b = parse(sys.argv[2]) if len(sys.argv) > 2 else None
dm = demangle(sorted(set(a) | set(b or {})))
print('%-52s %s' % ('kernel', 'vgpr scratch lds valu salu' + ('   |   second listing' if b else '')))
for k in sorted(set(a) | set(b or {}), key=lambda k: dm[k]):
    if len(sys.argv) > 3 and sys.argv[3] not in dm[k]:
        continue
    fa = ' '.join(str(v).rjust(6) for v in a.get(k, ['-'] * 5))
    fb = ('   | ' + ' '.join(str(v).rjust(6) for v in b.get(k, ['-'] * 5))) if b else ''
    print('%-52s %s%s' % (dm[k][-52:], fa, fb))
