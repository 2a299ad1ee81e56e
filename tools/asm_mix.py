#!/usr/bin/env python3
"""Instruction-mix summary of one kernel in a hipcc -S listing (dev tool).
usage: asm_mix.py k.s <substring of mangled kernel name>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(_Z\w*%s\w*):.*?^\s*s_endpgm' % re.escape(pat), s, flags=re.M | re.S)
body = m.group(0)
ins = [l.strip().split()[0] for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = collections.Counter(ins)
print(m.group(1), 'total instructions', sum(c.values()))
groups = collections.Counter()
for k, v in c.items():
    if re.match(r'v_(fma|mul|add|fmac)_f64', k): groups['f64 arith'] += v
    elif k.startswith(('v_rcp', 'v_div')): groups[k] += v
    elif k.startswith('ds_'): groups[k] += v
    elif k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): groups[k] += v
    elif k.startswith('s_waitcnt'): groups['s_waitcnt'] += v
    elif k.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): groups[k] += v
    elif k.startswith('s_'): groups['salu/branch'] += v
    elif k.startswith('v_cndmask'): groups['v_cndmask'] += v
    elif k.startswith('v_'): groups['other valu'] += v
    else: groups[k] += v
for k, v in sorted(groups.items(), key=lambda x: -x[1]):
    print('  %-28s %d' % (k, v))
print(c.most_common(30))
for key in ('.vgpr_count', '.sgpr_count', '.vgpr_spill_count', '.sgpr_spill_count', '.private_segment_fixed_size'):
    mm = re.search(r'%s:\s*(\d+)[^\n]*\n(?:[^\n]*\n){0,40}?[^\n]*%s' % (re.escape(key), re.escape(pat)), s)
