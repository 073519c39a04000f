#!/usr/bin/env python3
"""Outline of a kernel in hipcc's device assembly: waits, barriers, memory instructions, MFMAs and branch labels in program order, runs
of the same instruction collapsed.   usage: python tools/asm_outline.py <dev.s> <kernel-name-substring> [first_line last_line]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and sys.argv[2] in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
keys = ('s_waitcnt', 's_barrier', 'buffer_load', 'global_load', 'v_mfma', 'ds_read', 'ds_write', 'global_store', 'buffer_store', 's_cbranch', 's_branch', 's_load', 'ds_add', 'global_atomic')
out, last, cnt = [], None, 0
for l in lines[start:end]:
    t = l.strip()
    k = None
    if re.match(r'\.LBB\S*:', t): k = t.split()[0]
    else:
        for kk in keys:
            if t.startswith(kk):
                op = t.split()[0]
                k = op + (' ' + t[len(op):].split(';')[0].strip() if kk in ('s_waitcnt', 's_cbranch', 's_branch') else (' lds' if t.rstrip().endswith(' lds') or ' lds ' in t else ''))
                break
    if k is None: continue
    if k == last: cnt += 1
    else:
        if last: out.append(f'{last} x{cnt}' if cnt > 1 else last)
        last, cnt = k, 1
out.append(f'{last} x{cnt}')
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(out)
print('\n'.join(out[lo:hi]))
