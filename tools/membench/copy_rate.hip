// libmembench.so - the yardstick bench.py prints beside the step kernel's roofline fraction: what this GPU's memory
// system gives a plain streaming copy and a plain fill (16 bytes per lane, grid-stride, far beyond the caches).
// Bench tooling, not part of libmgrid_hip.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = src[i];
}
__global__ __launch_bounds__(256) void fill16(uint4 *__restrict__ dst, size_t n16, uint32_t v) {
    const uint4 w = make_uint4(v, v, v, v);
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = w;
}
}  // namespace

// Times `iters` launches (after 3 untimed ones) with HIP events on `stream`; returns the milliseconds of all of them,
// or a negative value on a HIP error.  `src` NULL: fill `dst` instead of copying.  Device pointers, 16-byte aligned.
extern "C" float mb_stream_ms(const void *src, void *dst, size_t bytes, int iters, int blocks, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n16 = bytes / 16;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.0f;
    for (int i = 0; i < iters + 3; ++i) {
        if (i == 3) (void)hipEventRecord(a, st);
        if (src)
            hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, st, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16);
        else
            hipLaunchKernelGGL(fill16, dim3(blocks), dim3(256), 0, st, static_cast<uint4 *>(dst), n16, 0x01010101u);
    }
    (void)hipEventRecord(b, st);
    float ms = -1.0f;
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) ms = -1.0f;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return hipGetLastError() == hipSuccess ? ms : -1.0f;
}
