#!/usr/bin/env python3
"""Run-length view of the instruction classes of one kernel in a gfx950 assembly file:
M mfma, v VALU, L LDS, G global/buffer, W s_waitcnt, B barrier, J branch, s other scalar.
usage: python tools/isa_seq.py file.s kernel-substring [first_mfma_index [count]]"""
import itertools
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*%s\w*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.end():s.find('s_endpgm', m.end())]
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]


def cls(l):
    op = l.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_')): return 'G'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith(('s_cbranch', 's_branch')): return 'J'
    if op.startswith('s_'): return 's'
    if op.startswith('v_'): return 'v'
    return '?'


seq = ''.join(cls(l) for l in lines)
idx = [i for i, c in enumerate(seq) if c == 'M']
a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 32
reg = seq[max(idx[a] - 60, 0):idx[min(a + n, len(idx) - 1)] + 10]
print(' '.join('%s%d' % (k, len(list(g))) if len(list(g2 := [k])) and (c := len(list(g))) > 1 else k for k, g in []) or
      ' '.join((k + str(c) if c > 1 else k) for k, c in ((k, len(list(g))) for k, g in itertools.groupby(reg))))
