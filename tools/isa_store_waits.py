"""Per kernel in a hipcc -S listing: number of global stores / atomics and how many of them have an
s_waitcnt vmcnt(0) as the closest preceding wait in their own basic block (gfx9 counts stores in
vmcnt, so that pattern serialises stores on each other's acknowledge)."""
import re
import sys

for path in sys.argv[1:]:
    name = None
    stats = {}
    pending_wait = False
    for l in open(path):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
            stats[name] = [0, 0, 0]
            pending_wait = False
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith('.LBB') or t.startswith('; %bb'):
            pending_wait = False
        elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
            pending_wait = True
            stats[name][2] += 1
        elif t.startswith('global_store') or t.startswith('global_atomic'):
            stats[name][0] += 1
            if pending_wait:
                stats[name][1] += 1
            pending_wait = False
        elif t.startswith('global_load') or t.startswith('buffer_'):
            pending_wait = False
        elif t.startswith('.Lfunc_end'):
            name = None
    for k, (st, w, tot) in stats.items():
        if st:
            print(f'{path.split("/")[-1]:18s} {k[:70]:70s} stores {st:4d}  waited {w:4d}  vmcnt0 {tot:4d}')
