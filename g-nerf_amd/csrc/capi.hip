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
