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

}  // namespace gnerf
