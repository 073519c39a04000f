"""The build's assembly pass (g-nerf_amd/csrc/pk_opsel_fixup.py) and the ISA lint of the built library (tools/isa_lint.py): no
packed-fp32 instruction of a kernel that runs v_mfma_f32_16x16x32_* may take the low half of its result from the high register of
src1 -- on MI355X that read returns 0.0 in lanes 48-63 now and then while another wave of the SIMD has the matrix instruction in
flight (profiles/r04_pk_opsel_hazard.md).  CPU-only: text in, text out, and a disassembly of the library hipcc cross-compiled."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd', 'csrc'), os.path.join(ROOT, 'tools'), os.path.join(ROOT, 'g-nerf_amd')]
import pk_opsel_fixup as FX


def _evaluate(line, regs):
    """(lo, hi) of a v_pk_{mul,add,fma}_f32 line on a register file {name: float32}: the VOP3P semantics the pass relies on --
    op_sel picks the register of a pair for the LOW result, op_sel_hi for the HIGH one, neg_lo / neg_hi negate that half's source."""
    m = FX.INSTR.match(line)
    op, (ops, mods, _) = m.group(2), FX.split_operands(m.group(3))
    n = len(ops) - 1
    sel, sel_hi = mods.get('op_sel', [0] * n), mods.get('op_sel_hi', [1] * n)
    neg_lo, neg_hi = mods.get('neg_lo', [0] * n), mods.get('neg_hi', [0] * n)

    def src(k, which, neg):
        text = ops[1 + k]
        if text.startswith('v['):
            a = int(text[2:text.index(':')])
            v = regs['v%d' % (a + which)]
        else:
            v = np.float32(float(text))                      # an inline constant feeds both halves
        return np.float32(-v) if neg else v
    out = []
    for which_of, negs in ((sel, neg_lo), (sel_hi, neg_hi)):
        s = [src(k, which_of[k], negs[k]) for k in range(n)]
        if op == 'v_pk_mul_f32': out.append(np.float32(s[0] * s[1]))
        elif op == 'v_pk_add_f32': out.append(np.float32(s[0] + s[1]))
        else: out.append(np.float32(np.float64(s[0]) * np.float64(s[1]) + np.float64(s[2])))
    return out


CASES = [
    ('\tv_pk_mul_f32 v[114:115], v[122:123], v[124:125] op_sel:[0,1]', '\tv_pk_mul_f32 v[114:115], v[124:125], v[122:123] op_sel:[1,0]'),
    ('\tv_pk_mul_f32 v[6:7], v[10:11], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]', '\tv_pk_mul_f32 v[6:7], v[14:15], v[10:11] op_sel:[1,0] op_sel_hi:[0,1]'),
    ('\tv_pk_add_f32 v[2:3], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]', '\tv_pk_add_f32 v[2:3], v[2:3], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]'),
    ('\tv_pk_add_f32 v[4:5], v[8:9], 1.0 op_sel:[0,1] neg_lo:[1,0] neg_hi:[1,0]', None),        # a constant has no high register, but the rewrite stays valid
    ('\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]', '\tv_pk_fma_f32 v[0:1], v[4:5], v[2:3], v[6:7] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]'),
    ('\tv_pk_mul_f32 v[8:9], v[2:3], v[4:5] op_sel:[0,1] ; a comment', None),
]


@pytest.mark.parametrize('before,after', CASES)
def test_fixup_moves_the_high_register_select_to_src0_and_keeps_the_value(before, after):
    new, what = FX.fix_line(before)
    assert what == 'fixed'
    if after is not None: assert new == after
    assert not FX.hazardous(FX.split_operands(FX.INSTR.match(new).group(3))[1])
    rng = np.random.default_rng(3)
    regs = {'v%d' % i: np.float32(rng.standard_normal()) for i in range(130)}
    a, b = _evaluate(before, regs), _evaluate(new, regs)
    assert [x.tobytes() for x in a] == [x.tobytes() for x in b]


def test_fixup_leaves_safe_forms_alone_and_reports_what_it_cannot_repair(tmp_path):
    for line in ('\tv_pk_mul_f32 v[240:241], v[172:173], v[240:241] op_sel:[1,0]', '\tv_pk_mul_f32 v[120:121], v[122:123], v[124:125] op_sel_hi:[1,0]',
                 '\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]', '\tv_pk_fma_f32 v[4:5], v[20:21], v[12:13], v[4:5] op_sel:[0,0,1]', '\tv_mul_f32_e32 v1, v2, v3', '\tv_pk_mul_f16 v1, v2, v3 op_sel:[0,1]'):
        assert FX.fix_line(line) == (line, 'ok')
    # the high select on BOTH sources: two scalar instructions, the half whose destination the other still reads second
    for line in ('\tv_pk_fma_f32 v[4:5], v[20:21], v[12:13], v[4:5] op_sel:[1,1,0] op_sel_hi:[1,0,1]', '\tv_pk_fma_f32 v[6:7], v[6:7], v[2:3], 0 op_sel:[1,1,0] op_sel_hi:[1,0,0]',
                 '\tv_pk_mul_f32 v[6:7], v[6:7], v[2:3] op_sel:[1,1] neg_lo:[1,0]', '\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[0,1]'):
        two, what = FX.fix_line(line)
        assert what == 'split' and len(two) == 2 and all('v_pk_' not in t for t in two)
        rng = np.random.default_rng(5)
        regs = {'v%d' % i: np.float32(rng.standard_normal()) for i in range(40)}
        want = _evaluate(line, regs)
        got = dict(regs)
        for t in two:                                                        # executed in order, on the same register file
            op, rest = t.split()[0], t.split(None, 1)[1]
            dst, *src = [x.strip() for x in rest.split(',')]
            val = [(-1.0 if x.startswith('-') else 1.0) * (float(got[x.lstrip('-')]) if x.lstrip('-') in got else float(x)) for x in src]
            val = [np.float32(v) for v in val]
            got[dst] = np.float32(val[0] * val[1]) if op.startswith('v_mul') else np.float32(val[0] + val[1]) if op.startswith('v_add') else np.float32(np.float64(val[0]) * np.float64(val[1]) + np.float64(val[2]))
        d0 = int(FX.split_operands(FX.INSTR.match(line).group(3))[0][0][2:].split(':')[0])
        assert [got['v%d' % d0].tobytes(), got['v%d' % (d0 + 1)].tobytes()] == [x.tobytes() for x in want], line
    assert FX.fix_line('\tv_pk_mul_f32 v[2:3], v[2:3], v[2:3] op_sel:[1,1] op_sel_hi:[0,0]')[1] == 'unfixable'        # each half's destination is read by the other
    # a file: an instruction that cannot be repaired stops the pass in ANY function (the partner wave can belong to another kernel)
    body = ['\t.type\tplain_kernel,@function', 'plain_kernel:', '\tv_pk_add_f32 v[8:9], v[8:9], v[2:3] op_sel:[0,1] clamp', '\ts_endpgm',
            '\t.type\tmatrix_kernel,@function', 'matrix_kernel:', '\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]',
            '\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]', '\ts_endpgm']
    p = tmp_path / 'a.s'
    p.write_text('\n'.join(body))
    assert FX.main([str(p)]) == 0
    text = p.read_text()
    assert 'v_pk_mul_f32 v[0:1], v[4:5], v[2:3] op_sel:[1,0]' in text
    assert 'v_pk_add_f32 v[8:9], v[2:3], v[8:9] op_sel:[1,0] clamp' in text                                 # a trailing `clamp` stays at the end
    assert FX.main(['--check', str(p)]) == 0
    before = p.read_text()
    assert FX.main([str(p)]) == 0 and p.read_text() == before                    # a second pass changes nothing
    for where in (2, 7):                                                         # ... in the kernel without the matrix instruction too
        bad = list(body)
        bad[where] = '\tv_pk_mul_f32 v[2:3], v[2:3], v[2:3] op_sel:[1,1] op_sel_hi:[0,0]'
        p.write_text('\n'.join(bad))
        assert FX.main([str(p)]) == 1 and FX.main(['--check', str(p)]) == 1


def _assemble(tmp_path, body, name='sample'):
    """gfx950 object of an assembly text, through the toolchain's own assembler (skips where it is absent)"""
    import subprocess
    import isa_lint
    clang = os.path.join(isa_lint.LLVM, 'clang')
    if not os.path.isfile(clang) or not os.path.isfile(os.path.join(isa_lint.LLVM, 'llvm-objdump')):
        pytest.skip('no gfx950 assembler / disassembler here')
    src, obj = tmp_path / (name + '.s'), tmp_path / (name + '.o')
    src.write_text(body)
    subprocess.run([clang, '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', str(src), '-o', str(obj)], check=True)
    return str(obj)


SAMPLE = """
    .text
    .globl hazard_kernel
    .type hazard_kernel,@function
hazard_kernel:
    v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]
    v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]
    v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]
    v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]
    v_pk_add_f32 v[0:1], v[2:3], 1.0 op_sel:[0,1] clamp
    v_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel:[0,1] neg_lo:[1,0]
    v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]
    v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,1] op_sel_hi:[0,0,1] neg_hi:[0,1,0]
    v_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]
    v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]
    s_endpgm
    .globl clean_kernel
    .type clean_kernel,@function
clean_kernel:
    v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]
    v_pk_add_f32 v[0:1], v[2:3], v[4:5]
    s_endpgm
"""


def test_word_decoder_reads_the_select_from_the_encoding_and_needs_no_syntax(tmp_path, monkeypatch):
    """tools/isa_lint.py decides from the ENCODED words (VOP3P field, opcode, OP_SEL bit 1), the build pass from the assembly TEXT.
    On an assembled sample both find the same four instructions; with the text parser blinded (a syntax change) the word decoder still
    finds them and the lint reports the disagreement as an error of its own."""
    import isa_lint
    obj = _assemble(tmp_path, SAMPLE)
    k = isa_lint.lint(obj)
    assert k['hazard_kernel']['pk_src1_hi'] == 4 and k['hazard_kernel']['text_pk_src1_hi'] == 4 and not k['hazard_kernel']['disagree']
    assert k['hazard_kernel']['mfma_16x16x32'] == 1 and k['hazard_kernel']['pk_mov_src1_hi'] == 1
    assert k['clean_kernel']['pk_src1_hi'] == 0 and not k['clean_kernel']['disagree'] and k['clean_kernel']['mfma_16x16x32'] == 0
    assert isa_lint.main([obj]) == 1                                            # (an error in a kernel is an error of the library)
    import re
    monkeypatch.setattr(isa_lint.FX, 'INSTR', re.compile(r'^(\s*)(v_packed_(?:mul|add|fma)_f32)\s+(.*?)\s*(;.*)?$'))     # "the mnemonics were renamed"
    k = isa_lint.lint(obj)
    assert k['hazard_kernel']['pk_src1_hi'] == 4 and k['hazard_kernel']['text_pk_src1_hi'] == 0
    assert len(k['hazard_kernel']['disagree']) == 4 and len(k['clean_kernel']['disagree']) == 2      # every packed-fp32 instruction the text reader no longer sees
    monkeypatch.undo()
    monkeypatch.setattr(isa_lint.FX, 'MOD', re.compile(r'\b(opsel|opsel_hi|neg_lo|neg_hi)=\[([01](?:,[01])*)\]'))        # "the modifiers are spelt differently"
    k = isa_lint.lint(obj)
    assert k['hazard_kernel']['pk_src1_hi'] == 4 and k['hazard_kernel']['text_pk_src1_hi'] == 0 and len(k['hazard_kernel']['disagree']) == 4


def test_opcode_table_of_the_lint_is_the_toolchains(tmp_path):
    """The decoder's opcode numbers against the assembler of the toolchain that builds the library: one instruction per mnemonic."""
    import subprocess
    import isa_lint
    ops = {'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]': ('pk_f32', 'v_pk_fma_f32'), 'v_pk_mul_f32 v[0:1], v[2:3], v[4:5]': ('pk_f32', 'v_pk_mul_f32'),
           'v_pk_add_f32 v[0:1], v[2:3], v[4:5]': ('pk_f32', 'v_pk_add_f32'), 'v_pk_mov_b32 v[0:1], v[2:3], v[4:5]': ('pk_mov', None),
           'v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]': ('mfma128', 'v_mfma_f32_16x16x32_f16'),
           'v_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]': ('mfma128', 'v_mfma_f32_16x16x32_bf16'),
           'v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]': ('mfma128', 'v_mfma_f32_32x32x16_f16'),
           'v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]': ('mfma128', 'v_mfma_f32_32x32x16_bf16'),
           'v_mfma_i32_16x16x64_i8 v[0:3], v[4:7], v[8:11], v[0:3]': ('mfma128', 'v_mfma_i32_16x16x64_i8'),
           'v_mfma_f32_16x16x32_fp8_fp8 v[0:3], v[4:5], v[8:9], v[0:3]': ('mfma16x16x32', 'fp8_fp8'),
           'v_mfma_f32_16x16x32_bf8_bf8 v[0:3], v[4:5], v[8:9], v[0:3]': ('mfma16x16x32', 'bf8_bf8'),
           'v_mfma_f32_16x16x16_f16 v[0:3], v[4:5], v[8:9], v[0:3]': (None, None), 'v_mfma_f32_16x16x4_f32 v[0:3], v4, v8, v[0:3]': (None, None),
           'v_pk_fma_f16 v0, v1, v2, v3': (None, None), 'v_fma_f32 v0, v1, v2, v3': (None, None)}
    obj = _assemble(tmp_path, '.text\n' + '\n'.join(ops) + '\n', 'ops')
    dis = subprocess.run([os.path.join(isa_lint.LLVM, 'llvm-objdump'), '-d', obj], check=True, capture_output=True, text=True).stdout
    words = [int(isa_lint.WORDS.search(ln).group(2).split()[0], 16) for ln in dis.split('\n') if isa_lint.WORDS.search(ln)]
    assert len(words) == len(ops)
    for (text, (kind, detail)), w0 in zip(ops.items(), words):
        got_kind, got = isa_lint.decode(w0)
        assert got_kind == kind, (text, hex(w0), got_kind)
        if kind == 'pk_f32': assert got == (detail, 0)
        elif kind in ('mfma128', 'mfma16x16x32'): assert got == detail


def test_built_library_has_no_such_instruction_anywhere():
    import gnerf_hip
    import isa_lint
    if not os.path.isfile(gnerf_hip.LIB_PATH) or not os.path.isfile(os.path.join(isa_lint.LLVM, 'llvm-objdump')):
        pytest.skip('libgnerf_hip.so is not built / no llvm-objdump')
    kernels = isa_lint.lint(gnerf_hip.LIB_PATH)
    with_mfma = [n for n, v in kernels.items() if v['mfma_16x16x32']]
    assert len(kernels) > 100 and len(with_mfma) >= 8 and any('render_bwd_tiles_kernel' in n for n in with_mfma)
    # decided from the encoded words; anywhere in the library (kernels of different streams or processes can share a SIMD too)
    bad = {n: v['first'] for n, v in kernels.items() if v['pk_src1_hi']}
    assert not bad, bad
    # the build pass's own text parser sees the same instructions as the word decoder: it is not blind on this toolchain's syntax
    blind = {n: v['disagree'] for n, v in kernels.items() if v['disagree']}
    assert not blind, blind
    assert not any(v['text_pk_src1_hi'] for v in kernels.values())
    assert sum(1 for v in kernels.values() if v['pk_mov_src1_hi']) == 0           # v_pk_mov_b32 with OP_SEL[1]: none today (it would be listed, not rewritten)
    # no kernel names the destination of a scalar load in front of the s_waitcnt lgkmcnt(0) that covers it (the kernels that issue
    # s_load from asm statements -- bin_accumulate_kernel -- are the ones the compiler cannot protect)
    early = {n: v['smem_read_before_wait'] for n, v in kernels.items() if v['smem_read_before_wait']}
    assert not early, early
    assert any('bin_accumulate_kernel' in n for n in kernels)


def test_scalar_load_tracker_flags_a_copy_in_front_of_the_wait():
    """The pattern that faulted in round 5: the register allocator coalesced `rcB = rcC` into the wait's operand and copied the
    registers of an s_load issued from an asm statement BEFORE the wait."""
    import isa_lint
    bad = isa_lint.SmemTracker()
    for line in ('s_load_dwordx8 s[52:59], s[52:53], 0x0', 'v_mul_f32_e32 v1, v2, v3', 's_mov_b64 s[36:37], s[52:53]', 's_waitcnt lgkmcnt(0)', 's_mov_b64 s[38:39], s[54:55]'):
        bad.feed(line)
    assert len(bad.hits) == 1 and 's52' in bad.hits[0]
    good = isa_lint.SmemTracker()
    for line in ('s_load_dwordx8 s[52:59], s[52:53], 0x0', 's_load_dwordx2 s[4:5], s[0:1], 0x10', 'ds_read_b32 v1, v2', 's_waitcnt lgkmcnt(1)',
                 'v_add_u32_e32 v1, s60, v1', 's_waitcnt vmcnt(0) lgkmcnt(0)', 's_add_u32 s4, s4, s52', 's_load_dword s7, s[4:5], 0x0', 's_endpgm'):
        good.feed(line)
    assert not good.hits
    # an lgkmcnt above zero does not cover a scalar load (they return out of order); a load whose ADDRESS is in flight is flagged too
    t = isa_lint.SmemTracker()
    for line in ('s_load_dwordx2 s[4:5], s[0:1], 0x0', 's_waitcnt lgkmcnt(1)', 's_load_dword s7, s[4:5], 0x0'):
        t.feed(line)
    assert len(t.hits) == 1
