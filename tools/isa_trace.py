#!/usr/bin/env python3
"""Compressed control-flow trace of one kernel of a hipcc -S listing: labels, branches, barriers, vmcnt waits,
and per-block counts of MFMA / LDS / memory / spill instructions.  python tools/isa_trace.py file.s name-substring"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, flags=re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    cnt = {}

    def flush():
        global cnt
        if cnt:
            print('      ', dict(cnt))
            cnt = {}

    for i, line in enumerate(m.group(2).split('\n')):
        t = line.strip()
        if not t or t.startswith(';'):
            continue
        op = t.split(' ')[0]
        if t.startswith('.LBB') or op.startswith(('s_cbranch', 's_branch')) or op == 's_barrier' or (
                op == 's_waitcnt' and 'vmcnt' in t):
            flush()
            print(i, t[:80])
        else:
            for k in ('v_mfma', 'scratch_load', 'scratch_store', 'ds_read', 'global_store', 'global_load_lds',
                      'global_load', 'v_exp', 'v_permlane'):
                if op.startswith(k):
                    cnt[k] = cnt.get(k, 0) + 1
                    break
    flush()
    break
