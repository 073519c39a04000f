#!/usr/bin/env python3
"""Assembly pass of the gfx950 build (compile_unit.sh runs it on the device assembly of every translation unit): no packed-fp32
instruction may take the LOW half of its result from the HIGH register of src1.

Why.  On MI355X a `v_pk_mul_f32 D, S0, S1 op_sel:[0,1]` (low result = S0.lo * S1.hi) issued by one wave while ANOTHER wave of the same
SIMD has a `v_mfma_f32_16x16x32_f16` (the gfx950 form with 128-bit A / B operands) in flight sometimes reads S1.hi as 0.0 in lanes
48-63.  tools/probes/pk_opsel_hazard_probe.hip shows it in isolation (profiles/r04_pk_opsel_hazard_probe.txt): v_pk_mul / v_pk_fma /
v_pk_add with that select are wrong in lanes 48-63 only and next to that matrix instruction only -- between 2e-6 and 23 % of the
executions, depending on how the two waves' instruction streams line up; every wrong result examined had read 0.0 -- while the same
select on src0 or on v_pk_fma's src2, op_sel_hi on src1, and every form next to v_mfma_f32_16x16x16_f16, v_mfma_f32_16x16x4_f32, plain
VALU work or an idle partner give 0 errors in 6.6e9 lane-results each.  In the renderer it cost the backward tile kernel's f16 form a
16-lane quarter of one product about once per 2 000 tiles (profiles/r04_pk_opsel_hazard.md: assembly-level
variants of that kernel -- operands exchanged or two v_mul_f32: exact; wait states, fresh operands, other neighbours: no change).
The compiler knows nothing of this and picks the src1 form whenever register allocation leaves a broadcast value in an odd register.

What.  Every `v_pk_{mul,add,fma}_f32` whose op_sel selects the high register for src1 has src0 and src1 exchanged (all three are
commutative in src0 / src1) together with their op_sel / op_sel_hi / neg_lo / neg_hi entries.  An instruction that selects the high
register on BOTH sources cannot be repaired this way and is written as two scalar instructions (v_mul / v_add / v_fma_f32: the same
IEEE operation per half), the half whose destination the other still reads going second.  Should neither order work, the instruction
stops the build -- in any kernel: the partner wave of the hazard can belong to ANOTHER kernel that shares the SIMD (a side stream, RCCL,
another process), so a kernel without a matrix instruction of its own is not exempt.  `--check` only reports (used by the tests on the
disassembly of the built library): exit status 1 if any such instruction is present anywhere.
"""
import re
import sys

INSTR = re.compile(r'^(\s*)(v_pk_(?:mul|add|fma)_f32)(?:_e64)?\s+(.*?)\s*(;.*|//.*)?$')
MOD = re.compile(r'\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01](?:,[01])*)\]')
FLAG = re.compile(r'(?:^|\s)(clamp)(?=\s|$)')          # operand-less trailing modifiers: taken off before the operands are split, re-appended after


def split_operands(text):
    """'v[1:2], v[3:4], 1.0 op_sel:[0,1]' -> (['v[1:2]', 'v[3:4]', '1.0'], {'op_sel': [0, 1]}, order of the modifiers)"""
    mods, order = {}, []
    def take(m):
        mods[m.group(1)] = [int(x) for x in m.group(2).split(',')]
        order.append(m.group(1))
        return ''
    ops = MOD.sub(take, text)
    flags = FLAG.findall(ops)
    if flags:
        ops = FLAG.sub('', ops)
        mods['_flags'] = flags
    depth, cur, out = 0, '', []
    for ch in ops:
        if ch == '[': depth += 1
        if ch == ']': depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur.strip()); cur = ''
        else:
            cur += ch
    if cur.strip(): out.append(cur.strip())
    return out, mods, order


def hazardous(mods):
    """the low half of the result reads the high register of src1.  (The same select on src0, on v_pk_fma_f32's src2, and op_sel_hi
    on any source are measured safe: tools/probes/pk_opsel_hazard_probe.hip.)"""
    sel = mods.get('op_sel')
    return bool(sel) and len(sel) >= 2 and sel[1] == 1


def _half(operand, which, neg):
    """the 32-bit operand text of one half of a packed source: register `which` (0 low, 1 high) of a pair, or the constant itself"""
    m = re.match(r'^([vs])\[(\d+):(\d+)\]$', operand)
    text = '%s%d' % (m.group(1), int(m.group(2)) + which) if m else operand
    return ('-' + text) if neg else text


def split_line(indent, op, ops, mods):
    """A packed instruction that selects the high register on BOTH of src0 and src1 cannot be repaired by an exchange: write it as two
    scalar instructions (same IEEE operation per half).  The half whose destination register the other half still has to read goes
    second; None if each half's destination is a source of the other."""
    n = len(ops) - 1
    sel, sel_hi = mods.get('op_sel', [0] * n), mods.get('op_sel_hi', [1] * n)
    neg_lo, neg_hi = mods.get('neg_lo', [0] * n), mods.get('neg_hi', [0] * n)
    scalar = {'v_pk_mul_f32': 'v_mul_f32_e64', 'v_pk_add_f32': 'v_add_f32_e64', 'v_pk_fma_f32': 'v_fma_f32'}[op]
    d = re.match(r'^v\[(\d+):(\d+)\]$', ops[0])
    if not d: return None
    dst = ['v%d' % int(d.group(1)), 'v%d' % (int(d.group(1)) + 1)]
    srcs = [[_half(ops[1 + k], sel[k], neg_lo[k]) for k in range(n)], [_half(ops[1 + k], sel_hi[k], neg_hi[k]) for k in range(n)]]
    reads = [set(x.lstrip('-') for x in srcs[h]) for h in (0, 1)]
    if dst[0] not in reads[1]: order = (0, 1)
    elif dst[1] not in reads[0]: order = (1, 0)
    else: return None
    tail = ''.join(' ' + f for f in mods.get('_flags', []))
    return ['%s%s %s, %s%s' % (indent, scalar, dst[h], ', '.join(srcs[h]), tail) for h in order]


def fix_line(line):
    """-> (new line or list of lines, 'ok' | 'fixed' | 'split' | 'unfixable')"""
    m = INSTR.match(line)
    if not m: return line, 'ok'
    indent, op, rest, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ''
    ops, mods, order = split_operands(rest)
    if not hazardous(mods): return line, 'ok'
    nsrc = len(ops) - 1
    if any(len(mods[name]) != nsrc for name in order): return line, 'unfixable'
    if mods['op_sel'][0] == 1:
        two = split_line(indent, op, ops, mods)
        return (two, 'split') if two else (line, 'unfixable')
    ops[1], ops[2] = ops[2], ops[1]
    for name in order:
        v = mods[name]
        v[0], v[1] = v[1], v[0]
    text = '%s%s %s' % (indent, op, ', '.join(ops))
    for name in order:
        v = mods[name]
        default = 1 if name == 'op_sel_hi' else 0
        if any(x != default for x in v): text += ' %s:[%s]' % (name, ','.join(str(x) for x in v))
    text += ''.join(' ' + f for f in mods.get('_flags', []))
    return (text + (' ' + comment if comment else '')).rstrip(), 'fixed'


def main(argv):
    check = '--check' in argv
    paths = [a for a in argv if not a.startswith('--')]
    if len(paths) != 1:
        print('usage: pk_opsel_fixup.py [--check] <assembly or disassembly>', file=sys.stderr)
        return 2
    lines = open(paths[0]).read().split('\n')
    # function of each line (compiler assembly: `.type name,@function` ... `name:`; disassembly: `<name>:`), and the functions that
    # contain the 128-bit matrix instruction
    func, owner, has_mfma = None, [None] * len(lines), set()
    for i, line in enumerate(lines):
        m = re.match(r'^\s*\.type\s+([\w.$]+),@function', line) or re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m: func = m.group(1)
        owner[i] = func
        if 'v_mfma_f32_16x16x32' in line: has_mfma.add(func)
    fixed = split = unfixable = tolerated = 0
    for i, line in enumerate(lines):
        if 'v_pk_' not in line: continue
        if check:
            m = INSTR.match(re.sub(r'^\s*[0-9a-f]+:\s*', '', re.sub(r'\s*//.*$', '', line)))       # objdump prefixes / suffixes
            if m and hazardous(split_operands(m.group(3))[1]):
                fixed += 1
                if owner[i] not in has_mfma: tolerated += 1
                if fixed <= 8: print('%s:%d: %s' % (paths[0], i + 1, line.strip()), file=sys.stderr)
            continue
        new, what = fix_line(line)
        if what == 'fixed': lines[i] = new; fixed += 1
        elif what == 'split': lines[i] = '\n'.join(new); split += 1
        elif what == 'unfixable':
            # an error in ANY kernel: the hazard is between two waves of a SIMD, and the other wave can belong to another kernel (a side
            # stream, RCCL, another process) -- a kernel without the matrix instruction of its own is not safe from it
            unfixable += 1
            if owner[i] not in has_mfma: tolerated += 1
            print('%s:%d: cannot move the high-register select off src1 (each half\'s destination is a source of the other: give the '
                  'result a register pair of its own): %s' % (paths[0], i + 1, line.strip()), file=sys.stderr)
    note = ' (%d of them in kernels without v_mfma_f32_16x16x32_* of their own)' % tolerated if tolerated else ''
    if check:
        print('[pk_opsel] %d packed-fp32 instruction(s) select the high register of src1%s' % (fixed, note))
        return 1 if fixed else 0
    if unfixable: return 1
    open(paths[0], 'w').write('\n'.join(lines))
    print('[pk_opsel] exchanged src0 / src1 of %d packed-fp32 instruction(s)%s%s' % (fixed, ', wrote %d as two scalar instructions' % split if split else '', note))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
