// Shared host-side helpers for libgnerf_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "gnerf_hip.h"

namespace gnerf {

// Thread-local message behind gnerf_last_error().
char* error_buffer();

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(GNERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return GNERF_OK;
}

inline hipStream_t as_stream(gnerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute is per device: remember which devices of this process have had a kernel's dynamic-LDS limit raised.  One static
// instance per kernel instantiation.  A device index outside the table (or a failing hipGetDevice) raises the limit on every call rather
// than sharing a slot with another device; two threads racing on the first call both raise it (idempotent).
struct PerDeviceOnce {
    bool done[64] = {};
    template <class K> int raise_lds(K kernel, const char* what, int bytes = 160 * 1024) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
        if (dev >= 0 && done[dev]) return GNERF_OK;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return fail(GNERF_E_LAUNCH, "%s: cannot raise the dynamic LDS limit", what);
        if (dev >= 0) done[dev] = true;
        return GNERF_OK;
    }
};

// Storage type -> arithmetic type (half computes in float, like the reference's InternalType).
template <class T> struct Arith { typedef float type; };
template <> struct Arith<double> { typedef double type; };

template <class T> __device__ __forceinline__ typename Arith<T>::type load_as(const T* p, int64_t i) {
    return static_cast<typename Arith<T>::type>(p[i]);
}
template <> __device__ __forceinline__ float load_as<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }

template <class T> __device__ __forceinline__ void store_as(T* p, int64_t i, typename Arith<T>::type v) { p[i] = static_cast<T>(v); }
template <> __device__ __forceinline__ void store_as<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

constexpr int kNumCU = 256;   // MI355X
constexpr int kNumXCD = 8;

// ---- the modulated convolution's epilogue on one vector of adjacent channels (shared by csrc/modconv.hip and the fused
// blur + epilogue of csrc/upfirdn2d.hip):
//   t = round_T(x * T(sc) + noise)        (SCALE / NOISE; the fp16 form is one packed half FMA / multiply per pair, see modconv.hip)
//   y = clamp(act(t + bias) * gain)       (ACT 1 linear, 3 lrelu)
//   NEXT: round_T(y) * nx                 (nx already rounded to T: the next layer's input scaling folded in)
template <class T, int VEC> struct alignas(sizeof(T) * VEC) Pk { T v[VEC]; };
template <class T> __device__ __forceinline__ float round_to(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__half>(float v) { return __half2float(__float2half(v)); }

// ACT: 1 linear, 3 lrelu, kActLrelu01 = lrelu whose slope the LAUNCHER has found in [0, 1] (the same bits as ACT 3 for such a slope:
// written with ACT 3 the test is a uniform branch in front of every element -- four scalar branches per 8-byte store in the fused
// convolution's epilogue, eight per vector in the blur's).
constexpr int kActLrelu01 = 13;
template <class T, int VEC, int ACT, bool SCALE, bool NOISE, bool NEXT>
__device__ __forceinline__ Pk<T, VEC> modconv_epilogue_vec(const Pk<T, VEC>& in, const float (&sc)[VEC], float nz, bool round_noise, const float (&bv)[VEC],
                                                          const float (&nx)[VEC], float alpha, float gain, float clamp) {
    float t[VEC];
    if constexpr (sizeof(T) == 2 && VEC % 2 == 0 && (SCALE || NOISE)) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            const h2 xv = {__builtin_bit_cast(_Float16, in.v[k]), __builtin_bit_cast(_Float16, in.v[k + 1])};
            const h2 s2 = {(_Float16)sc[k], (_Float16)sc[k + 1]};
            h2 r;
            if constexpr (NOISE) {
                if (round_noise || SCALE) {
                    const h2 n2 = {(_Float16)nz, (_Float16)nz};
                    r = __builtin_elementwise_fma(xv, s2, n2);
                } else {
                    r = (h2){(_Float16)((float)xv[0] + nz), (_Float16)((float)xv[1] + nz)};
                }
            } else {
                r = xv * s2;
            }
            t[k] = (float)r[0];
            t[k + 1] = (float)r[1];
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            float uu = float(load_as<T>(in.v, k));
            if constexpr (SCALE || NOISE) uu = round_to<T>(fmaf(uu, SCALE ? round_to<T>(sc[k]) : 1.f, NOISE ? ((round_noise || SCALE) ? round_to<T>(nz) : nz) : 0.f));
            t[k] = uu;
        }
    }
    Pk<T, VEC> out;
    // lrelu with a slope in (0, 1] is max(u, u * alpha) -- the same bits as the select for every u (signed zeros and NaN included; at
    // alpha == 0 the two differ for u = -inf only: the select gives -inf * 0 = NaN, the max -inf), one full-rate and one half-rate
    // instruction instead of one and two (tools/probes/valu_issue_probe.hip: v_cmp / v_cndmask cost 4.4 SIMD cycles each).  The clamp
    // is v_med3_f32, which sends NaN to -clamp exactly like the reference's kernel (bias_act.cu:143: `(y > -clamp & y < clamp) ? y :
    // (y >= 0) ? clamp : -clamp`), one instruction instead of four -- so with a clamp a NaN input leaves a GPU epilogue as -clamp, where
    // the PyTorch-op forms (CPU tensors; x.clamp) keep the NaN: include/gnerf_hip.h states it at gnerf_bias_act.
    const bool slope01 = alpha >= 0.f && alpha <= 1.f;
#pragma unroll
    for (int k = 0; k < VEC; k++) {
        const float uu = t[k] + bv[k];
        float r = uu;
        if (ACT == 3) r = slope01 ? fmaxf(uu, uu * alpha) : (uu > 0.f ? uu : uu * alpha);          // lrelu
        if (ACT == kActLrelu01) r = fmaxf(uu, uu * alpha);                                       // lrelu, slope checked by the launcher
        r *= gain;
        if (clamp >= 0.f) r = __builtin_amdgcn_fmed3f(r, -clamp, clamp);
        if constexpr (sizeof(T) == 2 && VEC % 2 == 0) {
            t[k] = r;                                                 // rounded and scaled in pairs below
        } else {
            if constexpr (NEXT) r = round_to<T>(r) * nx[k];
            store_as<T>(out.v, k, r);
        }
    }
    if constexpr (sizeof(T) == 2 && VEC % 2 == 0) {
        // fp16: round two values with one v_cvt_pk_f16_f32 (round to nearest even, as the single conversion) and scale the pair with one
        // v_pk_mul_f16 -- the same bits as element by element (nx holds fp16 values), 1.5 instructions per pair instead of 4
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        struct alignas(sizeof(T) * VEC) Words { unsigned w[VEC / 2]; } packed;
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            h2 rr = {(_Float16)t[k], (_Float16)t[k + 1]};
            if constexpr (NEXT) rr = rr * (h2){(_Float16)nx[k], (_Float16)nx[k + 1]};
            packed.w[k / 2] = __builtin_bit_cast(unsigned, rr);
        }
        return __builtin_bit_cast(Pk<T, VEC>, packed);
    }
    return out;
}

}  // namespace gnerf
