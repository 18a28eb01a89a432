// How fast can an MI355X take a radix pass's write pattern, with nothing else going on?  (tools/, not part of the library)
// Every "tile" writes RUN consecutive keys to each of 512 streams (one per digit); streams are n / 512 keys apart.
// build: hipcc -O3 --offload-arch=gfx950 tools/scatter_bench.hip -o gpurun_out/scatter_bench ; run: gpurun_out/scatter_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
template <int RUN>
__global__ __launch_bounds__(512) void scatter(u64* out, u64 stride, u32 tiles, u32* counter) {
    __shared__ u32 tk;
    for (;;) {
        if (threadIdx.x == 0) tk = atomicAdd(counter, 1u);
        __syncthreads();
        const u32 t = tk;
        __syncthreads();
        if (t >= tiles) break;
#pragma unroll
        for (int i = 0; i < RUN; i++) {          // RUN * 512 keys per tile: slot s -> digit s / RUN, place s % RUN
            const u32 s = (u32)i * 512 + threadIdx.x;
            const u32 d = s / RUN, r = s % RUN;
            out[(u64)d * stride + (u64)t * RUN + r] = ((u64)t << 20) | s;
        }
    }
}
template <int RUN>
static void run(u64* out, u64 n, u32* counter) {
    const u64 stride = n / 512;
    const u32 tiles = (u32)(stride / RUN);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(counter, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(scatter<RUN>, dim3(512), dim3(512), 0, 0, out, stride, tiles, counter);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("run of %3d keys (%4d B): %7.2f ms for %.1f GB = %.2f TB/s\n", RUN, RUN * 8, ms, 8.0 * stride * 512 / 1e9, 8.0 * stride * 512 / 1e9 / ms);
    }
}
int main() {
    const u64 n = 6221650873ull;
    u64* out; u32* counter;
    if (hipMalloc(&out, 8 * n) != hipSuccess || hipMalloc(&counter, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<16>(out, n, counter); run<32>(out, n, counter); run<64>(out, n, counter); run<128>(out, n, counter);
    return 0;
}
