// Memory-only stand-in for one step of the 131072 x 8 workload: per wavefront the same bytes in and
// out (16 environments: 624 B read, 1344 B written each), no computation.  Gives the floor that the
// HBM system sets for the step kernel's access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void traffic(const uint4 *__restrict__ in, uint4 *__restrict__ out, int rd16, int wr16) {
    const int lane = threadIdx.x;
    const uint4 *src = in + static_cast<size_t>(blockIdx.x) * rd16;
    uint4 *dst = out + static_cast<size_t>(blockIdx.x) * wr16;
    uint4 acc = make_uint4(lane, 0, 0, 0);
    for (int i = lane; i < rd16; i += 64) {
        const uint4 v = src[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    for (int i = lane; i < wr16; i += 64) dst[i] = acc;
}

int main(int argc, char **argv) {
    const int envs = 131072, per = argc > 1 ? atoi(argv[1]) : 16;
    const int blocks = envs / per;
    const int rd16 = per * 624 / 16, wr16 = per * 1344 / 16;
    uint4 *in, *out;
    hipMalloc(&in, static_cast<size_t>(blocks) * rd16 * 16);
    hipMalloc(&out, static_cast<size_t>(blocks) * wr16 * 16);
    hipMemset(in, 1, static_cast<size_t>(blocks) * rd16 * 16);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) traffic<<<blocks, 64>>>(in, out, rd16, wr16);
    hipEventRecord(a);
    const int n = 200;
    for (int i = 0; i < n; ++i) traffic<<<blocks, 64>>>(in, out, rd16, wr16);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / n, bytes = static_cast<double>(envs) * (624 + 1344);
    printf("envs/wave %d: %.1f us per launch, %.2f TB/s (%.0f MB)\n", per, us, bytes / us / 1e6, bytes / 1e6);
    return 0;
}
