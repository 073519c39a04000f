# Sourced by build.sh and tools/build_variants.sh.
LLVM="${LLVM_BIN:-/opt/rocm/lib/llvm/bin}"
# Every translation unit goes through its device ASSEMBLY: pk_opsel_fixup.py exchanges src0 / src1 of each packed-fp32 instruction
# that takes the low half of its result from the high register of src1 -- a form that, on MI355X, sometimes reads 0.0 in lanes 48-63
# while another wave of the SIMD runs v_mfma_f32_16x16x32_f16 (the script's header has the measurements; tools/isa_lint.py checks the
# built library).  The device side is then assembled, linked and bundled with the toolchain's own tools and handed to the host
# compile; the instruction streams are otherwise the compiler's.
# compile_unit <name> <object> [extra flags]   (expects $here = csrc, $HIPCC, $FLAGS)
compile_unit() {
    local src="$1" obj="$2" extra="${3:-}" t="$here/.build_$(basename "$2" .o)"
    rm -rf "$t" && mkdir -p "$t"
    $HIPCC $FLAGS $extra --cuda-device-only -S "$here/$src.hip" -o "$t/dev.s"
    python3 "$here/pk_opsel_fixup.py" "$t/dev.s" > "$t/fixup.log"
    sed "s/^/[build] $src.hip: /" "$t/fixup.log"
    "$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$t/dev.s" -o "$t/dev.o"
    "$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared "$t/dev.o" -o "$t/dev.co"
    "$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
        -input=/dev/null -input="$t/dev.co" -output="$t/dev.hipfb"
    $HIPCC $FLAGS $extra --cuda-host-only -c "$here/$src.hip" -Xclang -fcuda-include-gpubinary -Xclang "$t/dev.hipfb" -o "$obj"
    rm -rf "$t"
}
