#!/usr/bin/env python3
"""dev tool: per-barrier-segment instruction census of one kernel in a hipcc -S listing.
usage: census_newton.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*%s\w*):.*?^\s*s_endpgm' % re.escape(sys.argv[2]), s, flags=re.M | re.S)
body = m.group(0).split('\n')
seg, cur = [], []
for l in body:
    t = l.strip()
    if not l.startswith('\t') or t.startswith(('.', ';')):
        continue
    op = t.split()[0]
    cur.append(op)
    if op == 's_barrier':
        seg.append(cur)
        cur = []
seg.append(cur)
for i, ops in enumerate(seg):
    c = collections.Counter(ops)
    tot = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))
    f64 = sum(v for k, v in c.items() if re.match(r'v_(fma|mul|add|fmac)_f64', k))
    print('%2d n %5d f64 %4d scratch ld/st %3d/%3d ds rd/wr %3d/%3d global %3d cndmask %3d div %3d rcp %3d' % (
        i, len(ops), f64, tot('scratch_load'), tot('scratch_store'), tot('ds_read'), tot('ds_write'), tot('global_'),
        tot('v_cndmask'), tot('v_div_'), tot('v_rcp')))
