// Library-level entry points of libgnerf_hip.so (error reporting, version).
#include "common.h"

namespace gnerf {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace gnerf

extern "C" int gnerf_abi_version(void) { return GNERF_ABI_VERSION; }
extern "C" const char* gnerf_last_error(void) { return gnerf::error_buffer(); }
extern "C" const char* gnerf_build_info(void) {
    return "libgnerf_hip gfx950 (CDNA4) hipcc " __VERSION__;
}

// ---- the shader clock under whatever else runs (measurement aid of bench.py: the roofline's "peak" is cycles per second, and the chip
// holds its clock well under the 2.4 GHz of the data sheet while the render kernel runs -- DVFS, MI355X_MICROARCH.md).  One wave on one CU
// reads the shader-cycle counter (s_memtime) and the 100 MHz reference counter (s_memrealtime) until `ticks` reference ticks have passed
// and reports both differences.  Launched on a SIDE stream while the measured kernels run on theirs.  The loop ends by the reference
// counter alone, which always advances.
namespace {
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long* out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    out[0] = __builtin_amdgcn_s_memtime() - c0;
    out[1] = r1 - r0;
}
}  // namespace

extern "C" int gnerf_clock_sample(unsigned long long* out, double microseconds, gnerf_stream_t stream) {
    using namespace gnerf;
    if (!out) return fail(GNERF_E_ARG, "clock_sample: null pointer");
    if (!(microseconds > 0) || microseconds > 1e6) return fail(GNERF_E_ARG, "clock_sample: duration must be in (0, 1 s]");
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, as_stream(stream), out, (unsigned long long)(microseconds * 100.0));
    return check_launch("clock_sample");
}
