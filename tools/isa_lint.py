#!/usr/bin/env python3
"""ISA lint of the built library: no packed-fp32 instruction may take the LOW half of its result from the HIGH register of src1 --
the form that reads 0.0 in lanes 48-63 now and then while another wave of the SIMD runs v_mfma_f32_16x16x32_f16
(g-nerf_amd/csrc/pk_opsel_fixup.py has the measurements; the build exchanges the sources of every such instruction).

Two readers that share NOTHING but the instruction boundaries llvm-objdump prints:
  * WORDS (decides): the encoded instruction words of every gfx950 code object bundled in the library.  VOP3P is recognised by its
    encoding field (bits 31:23 = 0x1A7), the three packed-fp32 opcodes by bits 22:16 (0x30 fma, 0x31 mul, 0x32 add; 0x33 = v_pk_mov_b32,
    reported apart), the select by OP_SEL bit 1 (bit 12 of the first word); the 128-bit-operand matrix instructions by their opcodes.
    No mnemonic, no operand text: a change of the assembler's or the disassembler's SYNTAX cannot hide an instruction from it.
  * TEXT (cross-check): the disassembly text through the regular expressions of the build's own pass (pk_opsel_fixup.INSTR / MOD) --
    what the pass itself would see.  A disagreement between the two readers on any instruction is an error of its own: it means the
    pass is blind (or this decoder is wrong), whichever way round.
The opcode numbers are the toolchain's: tests/test_isa_cpu.py assembles one instruction of each kind and holds this table to it.
Round 5 adds a second check on the same disassembly: no instruction may name the destination of a scalar load in front of the
s_waitcnt lgkmcnt(0) that covers it (class SmemTracker: what an s_load inside an asm statement can lead the register allocator into).
usage: tools/isa_lint.py [library]      exit status 1 on an error; --json for one JSON line"""
import json, os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'g-nerf_amd', 'csrc'))
import pk_opsel_fixup as FX
LLVM = os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'

VOP3P_ENC = 0x1A7                                               # bits 31:23 of the first word (gfx9 / CDNA VOP3P, incl. the MAI ops)
PK_F32 = {0x30: 'v_pk_fma_f32', 0x31: 'v_pk_mul_f32', 0x32: 'v_pk_add_f32'}
PK_MOV = 0x33                                                   # v_pk_mov_b32: low result reads src0 only, listed when OP_SEL[1] is set
# matrix instructions whose A / B operands are 128 bits wide (the partner the hazard was measured with is the first one)
MFMA_128 = {0x54: 'v_mfma_f32_16x16x32_f16', 0x35: 'v_mfma_f32_16x16x32_bf16', 0x55: 'v_mfma_f32_32x32x16_f16', 0x37: 'v_mfma_f32_32x32x16_bf16',
            0x36: 'v_mfma_i32_16x16x64_i8'}
# the 64-bit-operand 16x16x32 forms (fp8 / bf8): counted with the others as "the kernel runs a 16x16x32 matrix instruction", as the text reader does
MFMA_16x16x32_OTHER = {0x70: 'bf8_bf8', 0x71: 'bf8_fp8', 0x72: 'fp8_bf8', 0x73: 'fp8_fp8'}
WORDS = re.compile(r'//\s*([0-9A-Fa-f]+):\s+((?:[0-9A-Fa-f]{8}\s*)+)$')


def decode(w0):
    """(kind, detail) of an instruction from its FIRST encoded word: ('pk_f32', (mnemonic, op_sel bits 2..0)), ('pk_mov', op_sel),
    ('mfma128', mnemonic), ('mfma16x16x32', name) or (None, None)."""
    if (w0 >> 23) != VOP3P_ENC:
        return None, None
    op, op_sel = (w0 >> 16) & 0x7F, (w0 >> 11) & 7
    if op in PK_F32: return 'pk_f32', (PK_F32[op], op_sel)
    if op == PK_MOV: return 'pk_mov', op_sel
    if op in MFMA_128: return 'mfma128', MFMA_128[op]
    if op in MFMA_16x16x32_OTHER: return 'mfma16x16x32', MFMA_16x16x32_OTHER[op]
    return None, None


def src1_high_for_low(op_sel):
    return bool(op_sel & 2)                                     # OP_SEL[1]: the low result takes the high register of src1


SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')


def sgprs(text):
    """the scalar registers an operand text names"""
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None: out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class SmemTracker:
    """Scalar loads complete out of order and only `s_waitcnt lgkmcnt(0)` covers them.  The compiler obeys that for its own loads; an
    s_load inside an `asm` statement is invisible to it, and a register copy the allocator places between the asm statement and the
    wait reads registers that have not landed (csrc/scatter_binned.inl met exactly that: a memory fault).  A linear scan per kernel:
    destinations of s_load / s_buffer_load stay `in flight` until an s_waitcnt whose lgkmcnt is 0; any instruction that names one of
    them meanwhile is reported.  (Straight-line approximation: the set is dropped at unconditional branches and at the program's end.)"""
    def __init__(self):
        self.inflight, self.hits = set(), []

    def feed(self, text):
        parts = text.split(None, 1)
        if not parts: return
        op, rest = parts[0], (parts[1] if len(parts) > 1 else '')
        if op == 's_waitcnt':
            if re.search(r'lgkmcnt\(0\)', rest) or re.fullmatch(r'0(x0)?', rest.strip()): self.inflight.clear()
            return
        if op in ('s_branch', 's_endpgm', 's_setpc_b64', 's_swappc_b64'):
            self.inflight.clear()
            return
        named = sgprs(rest)
        if op.startswith('s_load_dword') or op.startswith('s_buffer_load_dword'):
            ops = [o.strip() for o in rest.split(',')]
            dst = sgprs(ops[0]) if ops else set()
            src = named - dst if len(ops) > 1 and not (sgprs(','.join(ops[1:])) & dst) else sgprs(','.join(ops[1:]))
            bad = src & self.inflight
            if bad and len(self.hits) < 4: self.hits.append('%s   // reads s%s of a scalar load still in flight' % (text[:100], sorted(bad)[0]))
            self.inflight |= dst
            return
        bad = named & self.inflight
        if bad and len(self.hits) < 4: self.hits.append('%s   // names s%s of a scalar load still in flight' % (text[:100], sorted(bad)[0]))


def code_objects(lib, tmp):
    """the gfx950 code objects of every offload bundle in the library's .hip_fatbin section (one bundle per translation unit), written
    into the directory `tmp`; a plain code object is returned as is"""
    fat = os.path.join(tmp, 'fat.bin')
    r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section=.hip_fatbin=' + fat, lib], capture_output=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return [lib]
    blob = open(fat, 'rb').read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from('<Q', blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in triple and size:
                path = os.path.join(tmp, 'co%d.co' % len(out))
                open(path, 'wb').write(blob[pos + off:pos + off + size])
                out.append(path)
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def lint(lib):
    """per kernel: {'pk_src1_hi', 'mfma_16x16x32', 'first'} decided from the encoded words, plus 'pk_mov_src1_hi' (listed), 'text_pk_src1_hi'
    (what the build pass's own parser sees) and 'disagree' (instructions on which the two readers differ)."""
    kernels = {}
    with tempfile.TemporaryDirectory() as tmp:
        dumps = [subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', co], check=True, capture_output=True, text=True).stdout
                 for co in code_objects(lib, tmp)]
    for dis in dumps:
        name = None
        for line in dis.split('\n'):
            m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
            if m:
                name = m.group(1)
                kernels.setdefault(name, {'pk_src1_hi': 0, 'pk_mov_src1_hi': 0, 'mfma_16x16x32': 0, 'first': None, 'text_pk_src1_hi': 0, 'disagree': [],
                                          'smem': SmemTracker()})
                continue
            if name is None: continue
            k = kernels[name]
            w = WORDS.search(line)
            by_words = None
            if w:
                kind, detail = decode(int(w.group(2).split()[0], 16))
                if kind in ('mfma128', 'mfma16x16x32'): k['mfma_16x16x32'] += 1 if (kind == 'mfma16x16x32' or '16x16x32' in detail) else 0
                elif kind == 'pk_mov' and src1_high_for_low(detail): k['pk_mov_src1_hi'] += 1
                elif kind == 'pk_f32':
                    by_words = src1_high_for_low(detail[1])
                    if by_words:
                        k['pk_src1_hi'] += 1
                        k['first'] = k['first'] or ('%s op_sel=%s  // %s' % (detail[0], format(detail[1], '03b')[::-1], w.group(2).strip()))
            # the text reader: exactly what the build's pass would match
            text = re.sub(r'\s*//.*$', '', line).strip()
            k['smem'].feed(text)
            t = FX.INSTR.match(text) if 'v_pk_' in line else None
            by_text = bool(t and FX.hazardous(FX.split_operands(t.group(3))[1])) if t else None
            if by_text: k['text_pk_src1_hi'] += 1
            if (by_words is None) != (by_text is None) or (by_words is not None and by_words != by_text):
                if len(k['disagree']) < 4: k['disagree'].append(line.strip()[:160])
    for v in kernels.values(): v['smem_read_before_wait'] = v.pop('smem').hits
    return kernels


def main(argv):
    args = [a for a in argv if not a.startswith('--')]
    lib = args[0] if args else os.path.join(ROOT, 'g-nerf_amd', 'gnerf_hip', 'libgnerf_hip.so')
    k = lint(lib)
    errors = {n: v for n, v in k.items() if v['pk_src1_hi']}                 # anywhere in the library: waves of other kernels share SIMDs too
    blind = {n: v['disagree'] for n, v in k.items() if v['disagree']}
    movs = {n: v['pk_mov_src1_hi'] for n, v in k.items() if v['pk_mov_src1_hi']}
    early = {n: v['smem_read_before_wait'] for n, v in k.items() if v['smem_read_before_wait']}
    if '--json' in argv:
        print(json.dumps({'library': os.path.basename(lib), 'kernels': len(k), 'with_mfma_16x16x32': sum(1 for v in k.values() if v['mfma_16x16x32']),
                          'errors': {n: v['pk_src1_hi'] for n, v in errors.items()}, 'readers_disagree': blind, 'v_pk_mov_b32_op_sel1': movs,
                          'scalar_load_read_before_wait': early}))
    else:
        print('%d kernels, %d with a 16x16x32 matrix instruction (decoded from the instruction words)' % (len(k), sum(1 for v in k.values() if v['mfma_16x16x32'])))
        for n, v in errors.items():
            print('ERROR %s: %d packed-fp32 instruction(s) select the high register of src1%s, e.g. %s' %
                  (n[:90], v['pk_src1_hi'], '' if v['mfma_16x16x32'] else ' (no 16x16x32 matrix instruction in this kernel)', v['first']))
        for n, v in blind.items(): print('ERROR %s: the word decoder and the build pass\'s text parser disagree on %s' % (n[:90], v))
        for n, v in early.items(): print('ERROR %s: a scalar load\'s destination is used in front of the s_waitcnt lgkmcnt(0) that covers it: %s' % (n[:90], v))
        for n, c in movs.items(): print('note  %s: %d v_pk_mov_b32 with OP_SEL[1] (its low result reads src0 only)' % (n[:90], c))
    return 1 if errors or blind or early else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
