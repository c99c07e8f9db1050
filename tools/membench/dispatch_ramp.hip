// How long does the GPU take to get a one-round launch going?  Kernels whose wavefronts do a fixed amount of nothing
// (s_sleep) with the LDS and register footprint of the step kernel, launched as W workgroups of one wavefront or as
// W / k workgroups of k wavefronts: the difference between launch duration and the sleep is what dispatch costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void idle(int sleeps, unsigned *out) {
    extern __shared__ unsigned lds[];
    const int lane = threadIdx.x;
    lds[lane] = lane;
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);  // 127 * 64 clocks each
    if (out && lds[(lane + 1) & 63] == 0xFFFFFFFFu) out[0] = 1;
}

template <int WAVES>
double run(int waves, int lds_per_wave, int sleeps, int n) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void *>(idle<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_per_wave * WAVES);
    for (int i = 0; i < 20; ++i) idle<WAVES><<<waves / WAVES, 64 * WAVES, lds_per_wave * WAVES>>>(sleeps, nullptr);
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) idle<WAVES><<<waves / WAVES, 64 * WAVES, lds_per_wave * WAVES>>>(sleeps, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / n;
}

int main() {
    const int n = 300;
    for (int lds : {1024, 16384, 32768}) {
        for (int waves : {512, 2048, 4096, 16384}) {
            for (int sleeps : {0, 2}) {
                printf("lds/wave %5d waves %5d sleeps %d (%.1f us of sleep at 2.4 GHz): 1 wave/WG %6.2f us", lds, waves, sleeps,
                       sleeps * 127 * 64 / 2400.0, run<1>(waves, lds, sleeps, n));
                printf("   2/WG %6.2f us", run<2>(waves, lds, sleeps, n));
                if (lds * 4 <= 65536) printf("   4/WG %6.2f us", run<4>(waves, lds, sleeps, n));
                printf("\n");
            }
        }
    }
    return 0;
}
