// Stand-alone demonstration of the hazard csrc/pk_opsel_fixup.py removes from the build:
// a packed-fp32 instruction that takes the LOW half of its result from the HIGH register of src1, issued while ANOTHER wave of the
// same SIMD runs v_mfma_f32_16x16x32_f16, sometimes reads that register as 0.0 in lanes 48-63.
//
// One workgroup of 512 threads per CU = two waves per SIMD: waves 0-3 ("partners") spin on one kind of instruction, waves 4-7
// ("checkers") execute ONE packed-fp32 form over and over on known operands and compare each half of the result with the product
// computed by plain v_mul_f32.  Output: for every (checker form, partner kind) the number of wrong results per 16-lane quarter, how
// many of the wrong values were exactly +-0, and the first few wrong results with their operands (what was read instead: 0.0).
// HW_ID.SIMD_ID of both roles is recorded so that "shared a SIMD" is measured, not assumed.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_opsel_hazard_probe.hip -o /tmp/pk_probe && /tmp/pk_probe [out.json [iterations]]
// Result on MI355X (profiles/r04_pk_opsel_hazard_probe.*): v_pk_mul / v_pk_fma / v_pk_add with the low half from src1's high register
// are wrong in lanes 48-63 only, next to v_mfma_f32_16x16x32_f16 only (2.6e5 - 3.2e5 lane-results of 6.6e9; every sampled one read
// 0.0); the same select on src0 or on v_pk_fma's src2, op_sel_hi on src1, and every form next to v_mfma_f32_16x16x16_f16,
// v_mfma_f32_16x16x4_f32, a v_fma_f32 stream or an idle partner: 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

enum Form { PK_MUL_SRC1_HI = 0, PK_MUL_SRC0_HI, PK_MUL_SRC1_BCAST_LO, PK_FMA_SRC1_HI, PK_FMA_SRC2_HI, PK_ADD_SRC1_HI, PK_MUL_PLAIN, N_FORMS };
enum Partner { P_MFMA_16x16x32_F16 = 0, P_MFMA_16x16x16_F16, P_MFMA_16x16x4_F32, P_VALU_FMA, P_IDLE, N_PARTNERS };
static const char* kFormName[N_FORMS] = {"v_pk_mul_f32 op_sel:[0,1] (low <- src1.hi)", "v_pk_mul_f32 op_sel:[1,0] (low <- src0.hi)",
                                          "v_pk_mul_f32 op_sel_hi:[1,0] (high <- src1.lo)", "v_pk_fma_f32 op_sel:[0,1,0] (low <- src1.hi)",
                                          "v_pk_fma_f32 op_sel:[0,0,1] (low <- src2.hi)", "v_pk_add_f32 op_sel:[0,1] (low <- src1.hi)",
                                          "v_pk_mul_f32 (no select)"};
static const char* kPartnerName[N_PARTNERS] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_16x16x4_f32", "v_fma_f32 stream", "idle (s_sleep)"};

struct Out {
    unsigned wrong[4];          // per 16-lane quarter: results that differ from the v_mul_f32 / v_fma_f32 reference
    unsigned wrong_zero[4];     // ... of which the wrong value was +-0
    unsigned wrong_hi_half;     // errors in the HIGH half of the packed result
    unsigned simd_pairs_shared; // checker waves whose partner wave (same index - 4) reported the same SIMD_ID
    unsigned checked;           // packed instructions executed per lane
    unsigned n_samples;         // the first wrong results of the run: what did the instruction read instead?
    float sample[16][6];        // lane, got, want, src0.lo, src1.lo, src1.hi
};

template <int FORM>
__device__ __forceinline__ f2 run_form(f2 a, f2 b, f2 c) {
    f2 d;
    if (FORM == PK_MUL_SRC1_HI)            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b));
    else if (FORM == PK_MUL_SRC0_HI)       asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b));
    else if (FORM == PK_MUL_SRC1_BCAST_LO) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b));
    else if (FORM == PK_FMA_SRC1_HI)       asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else if (FORM == PK_FMA_SRC2_HI)       asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else if (FORM == PK_ADD_SRC1_HI)       asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b));
    else                                   asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
template <int FORM>
__device__ __forceinline__ f2 reference(f2 a, f2 b, f2 c) {       // the same values from scalar instructions, every operation rounded on its own
    switch (FORM) {
    case PK_MUL_SRC1_HI:       return (f2){__fmul_rn(a[0], b[1]), __fmul_rn(a[1], b[1])};
    case PK_MUL_SRC0_HI:       return (f2){__fmul_rn(a[1], b[0]), __fmul_rn(a[1], b[1])};
    case PK_MUL_SRC1_BCAST_LO: return (f2){__fmul_rn(a[0], b[0]), __fmul_rn(a[1], b[0])};
    case PK_FMA_SRC1_HI:       return (f2){__fmaf_rn(a[0], b[1], c[0]), __fmaf_rn(a[1], b[1], c[1])};
    case PK_FMA_SRC2_HI:       return (f2){__fmaf_rn(a[0], b[0], c[1]), __fmaf_rn(a[1], b[1], c[1])};
    case PK_ADD_SRC1_HI:       return (f2){__fadd_rn(a[0], b[1]), __fadd_rn(a[1], b[1])};
    default:                   return (f2){__fmul_rn(a[0], b[0]), __fmul_rn(a[1], b[1])};
    }
}

__device__ __forceinline__ unsigned simd_id() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    return (id >> 4) & 3;                                          // HW_ID[5:4] = SIMD_ID on gfx9
}

template <int FORM>
__global__ __launch_bounds__(512) void probe_kernel(int partner, int iters, Out* out, float* sink) {
    __shared__ unsigned simd_of[8];
    __shared__ int partners_done;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (lane == 0) simd_of[wv] = simd_id();
    if (tid == 0) partners_done = 0;
    __syncthreads();
    if (wv < 4) {
        // ---- partner: keep the SIMD's other issue slots busy with one instruction kind until the checkers are done (bounded)
        v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        h8 a8, b8;
        for (int k = 0; k < 8; k++) { a8[k] = (_Float16)(0.001f * (lane + k)); b8[k] = (_Float16)(0.002f * (lane - k)); }
        const h4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
        float x = 1.f + lane, y = 0.5f;
        for (int it = 0; it < iters * 4; it++) {
            if (partner == P_MFMA_16x16x32_F16) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0);
            } else if (partner == P_MFMA_16x16x16_F16) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0);
            } else if (partner == P_MFMA_16x16x4_F32) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[q], 0, 0, 0);
            } else if (partner == P_VALU_FMA) {
#pragma unroll
                for (int q = 0; q < 16; q++) { x = __fmaf_rn(x, 0.999f, y); asm volatile("" : "+v"(x)); }
            } else {
                __builtin_amdgcn_s_sleep(8);
            }
            if ((it & 63) == 63 && __atomic_load_n(&partners_done, __ATOMIC_RELAXED) >= 4) break;
        }
        sink[blockIdx.x * 512 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x;
        return;
    }
    // ---- checker
    unsigned wrong = 0, wrong_zero = 0, wrong_hi = 0;
    float l_got = 0, l_want = 0, l_a0 = 0, l_b0 = 0, l_b1 = 0;
    for (int it = 0; it < iters; it++) {
        const float s = 1.f + 0.001f * float((it * 64 + lane) & 1023);
        f2 a = {1.25f * s, -0.75f * s}, b = {3.f + 0.5f * lane, 0.375f + s}, c = {0.125f * s, -2.5f * s};
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        const f2 d = run_form<FORM>(a, b, c);
        const f2 r = reference<FORM>(a, b, c);
        if (__float_as_uint(d[0]) != __float_as_uint(r[0])) {
            wrong++; wrong_zero += (__float_as_uint(d[0]) << 1) == 0;
            // remember this lane's last wrong result in registers (an atomic in the loop changes its timing: ~100x fewer hits)
            l_got = d[0]; l_want = r[0]; l_a0 = a[0]; l_b0 = b[0]; l_b1 = b[1];
        }
        if (__float_as_uint(d[1]) != __float_as_uint(r[1])) { wrong_hi++; }
    }
    if (wrong) {
        atomicAdd(&out->wrong[lane >> 4], wrong); atomicAdd(&out->wrong_zero[lane >> 4], wrong_zero);
        const unsigned k = atomicAdd(&out->n_samples, 1u);
        if (k < 16) { float* q = out->sample[k]; q[0] = float(lane); q[1] = l_got; q[2] = l_want; q[3] = l_a0; q[4] = l_b0; q[5] = l_b1; }
    }
    if (wrong_hi) atomicAdd(&out->wrong_hi_half, wrong_hi);
    if (lane == 0) {
        if (simd_of[wv] == simd_of[wv - 4]) atomicAdd(&out->simd_pairs_shared, 1u);
        __atomic_fetch_add(&partners_done, 1, __ATOMIC_RELAXED);
    }
    if (tid == 256 && blockIdx.x == 0) out->checked = unsigned(iters);
}

template <int FORM>
static void launch(int partner, int iters, int blocks, Out* d_out, float* d_sink) {
    hipLaunchKernelGGL(probe_kernel<FORM>, dim3(blocks), dim3(512), 0, 0, partner, iters, d_out, d_sink);
}

int main(int argc, char** argv) {
    const int blocks = 256, iters = argc > 2 ? atoi(argv[2]) : 200000;
    Out* d_out; float* d_sink;
    (void)hipMalloc(&d_out, sizeof(Out)); (void)hipMalloc(&d_sink, sizeof(float) * blocks * 512);
    FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "[\n");
    bool first = true;
    for (int f = 0; f < N_FORMS; f++) {
        for (int p = 0; p < N_PARTNERS; p++) {
            (void)hipMemset(d_out, 0, sizeof(Out));
            switch (f) {
            case 0: launch<0>(p, iters, blocks, d_out, d_sink); break;
            case 1: launch<1>(p, iters, blocks, d_out, d_sink); break;
            case 2: launch<2>(p, iters, blocks, d_out, d_sink); break;
            case 3: launch<3>(p, iters, blocks, d_out, d_sink); break;
            case 4: launch<4>(p, iters, blocks, d_out, d_sink); break;
            case 5: launch<5>(p, iters, blocks, d_out, d_sink); break;
            default: launch<6>(p, iters, blocks, d_out, d_sink); break;
            }
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
            Out o;
            (void)hipMemcpy(&o, d_out, sizeof(Out), hipMemcpyDeviceToHost);
            const double total = double(iters) * blocks * 4 * 64;      // packed instructions x lanes
            printf("%-50s | %-26s | wrong low halves by lane quarter: %8u %8u %8u %8u  (zeros: %u %u %u %u)  high halves wrong: %u  of %.3g lane-results; %u of %d checker waves shared their partner's SIMD\n",
                   kFormName[f], kPartnerName[p], o.wrong[0], o.wrong[1], o.wrong[2], o.wrong[3], o.wrong_zero[0], o.wrong_zero[1], o.wrong_zero[2], o.wrong_zero[3],
                   o.wrong_hi_half, total, o.simd_pairs_shared, blocks * 4);
            for (unsigned k = 0; k < (o.n_samples < 6 ? o.n_samples : 6); k++)
                printf("    lane %2.0f: got %.9g, want %.9g; src0.lo %.9g, src1.lo %.9g, src1.hi %.9g -> the factor / addend the instruction used in place of src1.hi: %.9g\n", o.sample[k][0], o.sample[k][1],
                       o.sample[k][2], o.sample[k][3], o.sample[k][4], o.sample[k][5], (f == 5) ? o.sample[k][1] - o.sample[k][3]                                            // v_pk_add: got - src0.lo
                       : (f == 3 || f == 4) ? (o.sample[k][1] - 0.1f * o.sample[k][3]) / o.sample[k][3]      // v_pk_fma: src2.lo = 0.125 s = 0.1 * src0.lo is the addend
                       : o.sample[k][1] / o.sample[k][3]);
            if (js) {
                fprintf(js, "%s {\"form\": \"%s\", \"partner\": \"%s\", \"wrong_low_by_lane_quarter\": [%u, %u, %u, %u], \"of_which_zero\": [%u, %u, %u, %u], \"wrong_high\": %u, "
                            "\"lane_results\": %.0f, \"checker_waves_sharing_partner_simd\": %u, \"checker_waves\": %d}",
                        first ? "" : ",\n", kFormName[f], kPartnerName[p], o.wrong[0], o.wrong[1], o.wrong[2], o.wrong[3], o.wrong_zero[0], o.wrong_zero[1], o.wrong_zero[2],
                        o.wrong_zero[3], o.wrong_hi_half, total, o.simd_pairs_shared, blocks * 4);
                first = false;
            }
        }
    }
    if (js) { fprintf(js, "\n]\n"); fclose(js); }
    return 0;
}
