"""Print the order of memory / MFMA / barrier instructions of one kernel in a hipcc -S listing.

usage: python tools/isa_order.py k.s 'conv2d_pipe_kernelILi3ELi1ELi8ELi4ELi2E'
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().split('\n')
    pat = sys.argv[2]
    start = next(i for i, l in enumerate(text) if pat in l and l.startswith('_Z') and ':' in l)
    end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
    keys = ('global_load', 'global_store', 'global_atomic', 'v_mfma', 's_waitcnt', 's_barrier', 'ds_write', 'ds_read',
            's_cbranch', 'scratch_')
    runs = []
    for l in text[start + 1:end]:
        l = l.strip()
        if l.startswith('.LBB'):
            runs.append([l.split(':')[0], 1])
            continue
        if not l or l[0] in ';.':
            continue
        op = l.split()[0]
        key = next((k for k in keys if op.startswith(k)), None)
        if key is None:
            continue
        if key in ('global_load', 's_cbranch'):
            key = op
        if runs and runs[-1][0] == key:
            runs[-1][1] += 1
        else:
            runs.append([key, 1])
    print(' '.join(f'{k}x{n}' if n > 1 else k for k, n in runs))


if __name__ == '__main__':
    main()
