// Float-atomic throughput by memory scope and footprint on a multi-XCD device: every 16-lane group adds into one random 64-byte line of
// a buffer (what the plane-gradient scatter issues).  hipcc --offload-arch=gfx950 -O2 tools/probes/atomic_scope_probe.hip -o /tmp/asp && /tmp/asp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SCOPE, bool PARTITION>
__global__ __launch_bounds__(256) void k(float* buf, unsigned lines, int iters) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    const unsigned grp = gid >> 4, l = gid & 15;
    float* base = buf;
    if (PARTITION) base += size_t(blockIdx.x % 8) * size_t(lines) * 16;          // one private copy per XCD (workgroups go round-robin over the XCDs)
    unsigned h = grp * 2654435761u + 12345u;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const unsigned line = (h >> 8) % lines;
        float* p = base + size_t(line) * 16 + l;
        if (SCOPE == 3) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // integer add
        else if (SCOPE == 4) *p = 1.0f;                                                                                                 // plain store, for scale
        else if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <int SCOPE, bool PARTITION>
void run(const char* name, float* buf, unsigned lines, size_t floats_total) {
    const int blocks = 256 * 16, iters = 256;
    hipMemset(buf, 0, floats_total * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SCOPE, PARTITION>), dim3(blocks), dim3(256), 0, 0, buf, lines, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SCOPE, PARTITION>), dim3(blocks), dim3(256), 0, 0, buf, lines, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // check: the sum of the buffer must equal the number of adds
    const double adds = double(blocks) * 256 * (iters + 8);
    float* h = (float*)malloc(floats_total * 4); hipMemcpy(h, buf, floats_total * 4, hipMemcpyDeviceToHost);
    double sum = 0; for (size_t i = 0; i < floats_total; i++) sum += h[i];
    free(h);
    const double reqs = double(blocks) * 256 / 16 * iters;
    printf("%-34s %8.3f ms  %7.1f M 64-byte requests/ms... %7.2f GB/s of payload; sum %.0f of %.0f %s\n", name, ms, reqs / ms / 1e3, reqs * 64 / ms / 1e6, sum, adds, sum == adds ? "OK" : (SCOPE >= 3 ? "(not a float sum)" : "MISMATCH"));
}
int main() {
    const size_t max_lines = 256u * 1024 * 1024 / 64;
    float* buf; hipMalloc(&buf, max_lines * 16 * 8 * 4);
    {
        const unsigned lines = 100u * 1024 * 1024 / 64;            // a 100 MB gradient buffer
        const size_t total = size_t(lines) * 16 * 8;
        run<1, false>("agent scope, one 100 MB buffer", buf, lines, total);
        run<0, false>("workgroup scope, one buffer", buf, lines, total);
        run<2, false>("system scope, one buffer", buf, lines, total);
        run<1, true>("agent scope, 100 MB copy per XCD", buf, lines, total);
    }
    {
        const unsigned lines = 100u * 1024 * 1024 / 64;
        run<3, false>("u32 add, agent scope, 100 MB", buf, lines, size_t(lines) * 16);
        run<4, false>("plain 64-byte stores, 100 MB", buf, lines, size_t(lines) * 16);
    }
    // the same random adds over smaller and larger buffers: where does the rate come from -- L2 (4 MB per XCD), Infinity Cache (256 MB), HBM?
    for (unsigned mb : {1u, 4u, 16u, 32u, 64u, 100u, 256u}) {
        const unsigned lines = mb * 1024 * 1024 / 64;
        char name[64]; snprintf(name, sizeof name, "agent scope, one %u MB buffer", mb);
        run<1, false>(name, buf, lines, size_t(lines) * 16);
        snprintf(name, sizeof name, "agent scope, %u MB, copy per XCD", mb);
        if (mb <= 32) run<1, true>(name, buf, lines, size_t(lines) * 16 * 8);
    }
    return 0;
}
