#!/usr/bin/env python3
"""Per-kernel instruction census of a hipcc -S listing: python tools/isa_summary.py file.s [name-substring]"""
import re
import sys
from collections import Counter

KEYS = ('v_mfma', 'ds_read', 'ds_write', 'global_load', 'global_store', 'scratch_', 's_waitcnt', 's_barrier',
        'buffer_', 'v_exp', 'v_log', 'v_permlane', 'v_cvt_pk', 's_cbranch', 'v_accvgpr')


def kernels(text):
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', text, flags=re.S | re.M):
        yield m.group(1), m.group(2)


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, body in kernels(text):
        if want not in name:
            continue
        c = Counter()
        n = 0
        for line in body.split('\n'):
            t = line.strip().split(' ')[0]
            if not t or t.startswith((';', '.')):
                continue
            n += 1
            if t.startswith(KEYS):
                c[t] += 1
        print(name, n, 'instructions')
        for k, v in sorted(c.items()):
            print('   ', k, v)


if __name__ == '__main__':
    main()
