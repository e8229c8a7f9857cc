#!/usr/bin/env python3
"""Per-kernel instruction census of a gfx950 assembly file (hipcc --offload-device-only -S):
usage: python tools/isa_stats.py file.s substring [substring ...]"""
import re
import sys

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\w*%s\w*):' % re.escape(name), s, re.M):
        start = m.end()
        end = s.find('.end_amdhsa_kernel', start)
        code = s[start:s.find('s_endpgm', start)]
        cnt = lambda p: len(re.findall(p, code))
        meta = s[s.rfind('.amdhsa_kernel', 0, end):end]
        g = lambda k: (re.search(re.escape(k) + r'\s+(\S+)', meta) or [None, None])[1]
        print(m.group(1)[:70])
        print('   mfma %d | ds_read b32 %d read2_b32 %d b64 %d b128 %d | ds_write %d | lds-dma %d | global_load %d | '
              'global_store %d | scratch %d | s_waitcnt %d | s_barrier %d | v_cndmask %d' % (
                  cnt(r'v_mfma'), cnt(r'ds_read_b32'), cnt(r'ds_read2\w*_b32'), cnt(r'ds_read\w*_b64'), cnt(r'ds_read_b128'),
                  cnt(r'ds_write'), cnt(r'global_load_lds|buffer_load\w+ .*lds'), cnt(r'global_load_dword'),
                  cnt(r'global_store|buffer_store'), cnt(r'scratch_'), cnt(r's_waitcnt'), cnt(r's_barrier'), cnt('v_cndmask')))
        print('   vgpr %s (accum offset %s) sgpr %s lds %s scratch %s' % (
            g('.amdhsa_next_free_vgpr'), g('.amdhsa_accum_offset'), g('.amdhsa_next_free_sgpr'),
            g('.amdhsa_group_segment_fixed_size'), g('.amdhsa_private_segment_fixed_size')))
