#!/usr/bin/env python3
"""Instruction mix / register use of one kernel in a hipcc -S listing: isa_stats.py file.s kernel_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(name) + r'\S*):', s, re.M)
i = m.start()
e = s.index('.amdhsa_kernel', i)
ins = []
for l in s[i:e].split('\n'):
    t = l.strip()
    if not l.startswith('\t') or not t or t.startswith(('.', ';')):
        continue
    ins.append(t.split()[0])
print(m.group(1)[:60], 'instructions:', len(ins))
print(Counter(ins).most_common(45))
k = s[e:e + 4000]
for key in ('next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'group_segment_fixed_size', 'accum_offset'):
    mm = re.search(key + r'\s+(\S+)', k)
    print(key, mm.group(1) if mm else None)
