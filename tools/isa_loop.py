#!/usr/bin/env python3
"""Print the memory / wait / barrier skeleton of the MFMA loop of one kernel in a .s file."""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
a = s.index(name + ':')
b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo, hi = labels[m.group(1)], i
        seg = body[lo:hi + 1]
        mf = [j for j, x in enumerate(seg) if 'v_mfma' in x]
        if mf:
            print('loop len', hi - lo, 'mfma', len(mf), 'at', mf[0], '..', mf[-1])
            for j, x in enumerate(seg):
                if any(t in x for t in ('s_waitcnt', 's_barrier', 'global_load', 'buffer_load', 's_cbranch', 'ds_write', 'scratch_')):
                    print('     ', j, x.strip())
