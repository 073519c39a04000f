/* The drop-in boundary from plain C: include/gnerf_hip.h must compile as C99 (no C++, no torch types), the library must link, and the
 * entry points that need no GPU must behave: version, build string, workspace size, argument checking with a readable error message.
 * Built and run by tests/test_host_cpu.py::test_c_abi_from_plain_c. */
#include <stdio.h>
#include <string.h>
#include "gnerf_hip.h"

int main(void) {
    gnerf_render_params p;
    memset(&p, 0, sizeof p);
    if (gnerf_abi_version() != GNERF_ABI_VERSION) { printf("abi %d != header %d\n", gnerf_abi_version(), GNERF_ABI_VERSION); return 1; }
    if (!strstr(gnerf_build_info(), "gfx950")) { printf("build info: %s\n", gnerf_build_info()); return 2; }
    if (gnerf_render_workspace_bytes() < 64) return 3;
    if (gnerf_render_forward(NULL, NULL) != GNERF_E_ARG) return 4;
    if (!strstr(gnerf_last_error(), "null")) { printf("last error: %s\n", gnerf_last_error()); return 5; }
    if (gnerf_render_forward(&p, NULL) != GNERF_E_ARG) return 6;                      /* all-zero params: rejected before any launch */
    if (gnerf_bias_act(NULL, NULL, NULL, NULL, NULL, NULL, GNERF_F32, 16, 0, 0, 0, 1, 0.f, 1.f, -1.f, NULL) == GNERF_OK) return 7;   /* null tensors */
    printf("ok abi=%d sizeof(gnerf_render_params)=%u %s\n", gnerf_abi_version(), (unsigned)sizeof p, gnerf_build_info());
    return 0;
}
