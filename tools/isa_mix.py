#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a -save-temps .s file.  usage: isa_mix.py file.s <substring of the mangled name>"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
name = m.group(1)
a = s.index(name + ':')
b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
print(name, len(body), 'lines')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i


def mix(seg):
    c = Counter()
    for x in seg:
        x = x.strip()
        if not x or x.startswith(';') or x.startswith('.'):
            continue
        op = x.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('v_readlane') or op.startswith('v_writelane'):
            c['lane'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('s_waitcnt'):
            c['wait'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_nop'):
            c['nop'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['ds'] += 1
        elif op.startswith('buffer_') or op.startswith('global_') or op.startswith('scratch_'):
            c['vmem'] += 1
        else:
            c[op] += 1
    return dict(c)


for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        lo, hi = labels[mm.group(1)], i
        c = mix(body[lo:hi + 1])
        if c.get('mfma'):
            print(mm.group(1), lo, hi, c)
print('whole', mix(body))
