import sys, re
src = open('render-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
# the block: from the label .LBB3_95 to the first tbuf store region
i0 = next(i for i, l in enumerate(src) if l.startswith('.LBB3_95:'))
def find(pat, start=i0, n=1):
    c = 0
    for i in range(start, start + 400):
        if pat in src[i]:
            c += 1
            if c == n: return i
    raise SystemExit('pattern not found: ' + pat)
NOP2 = ['\ts_nop 7', '\ts_nop 7']
def variant(name):
    s = list(src)
    def ins(i, lines): s[i:i] = lines
    if name == 'v0': pass
    elif name == 'v1':
        i = find('v_pk_mul_f32 v[114:115], v[122:123], v[124:125] op_sel:[0,1]'); s[i] = '\tv_pk_mul_f32 v[114:115], v[124:125], v[122:123] op_sel:[1,0]'
    elif name == 'v2': ins(find('v_exp_f32_e32 v115, v115'), NOP2)
    elif name == 'v3': ins(find('v_add_f32_e32 v115, 1.0, v115'), NOP2)
    elif name == 'v4': ins(find('ds_read_b128 v[124:127], v234 offset:42720') + 1, ['\ts_waitcnt lgkmcnt(0)'] + NOP2)
    elif name == 'v5': ins(find('v_pk_mul_f32 v[120:121], v[116:117], s[96:97]'), NOP2)
    elif name == 'v6': ins(find('v_pk_mul_f32 v[114:115], v[122:123], v[124:125] op_sel:[0,1]'), NOP2)
    elif name == 'v7': ins(find('v_rcp_f32_e32 v116, v115') + 1, NOP2)
    elif name == 'v8': ins(find('v_rcp_f32_e32 v116, v115'), NOP2)        # between E and F
    else: raise SystemExit('unknown ' + name)
    open('dev_%s.s' % name, 'w').write('\n'.join(s))
for n in sys.argv[1:]: variant(n)
