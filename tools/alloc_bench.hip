// How long until device memory is usable, by API?  (tools/, not part of the library)  Every allocation is written once (first touch)
// before it is timed as "ready", freed, and taken again.
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/alloc_bench.hip -o gpurun_out/alloc_bench && gpurun_out/alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(uint4* p, size_t n16) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4); }
static double touch(void* p, size_t bytes) { double t0 = now(); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint4*)p, bytes / 16); hipDeviceSynchronize(); return (now() - t0) * 1e3; }
int main() {
    hipFree(0);
    const size_t GB = 1ull << 30;
    { void* w; hipMalloc(&w, GB); touch(w, GB); hipFree(w); }          // first launch out of the way
    for (int round = 0; round < 2; round++)
        for (size_t g : {8ull, 64ull}) {
            void* p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, g * GB);
            double t1 = now();
            if (e != hipSuccess) { printf("hipMalloc %zu GB failed\n", g); continue; }
            const double tt = touch(p, g * GB), tt2 = touch(p, g * GB);
            double t2 = now();
            hipFree(p);
            double t3 = now();
            printf("round %d hipMalloc      %3zu GB: alloc %8.1f ms (%5.1f ms/GB), first write %7.1f ms, second write %7.1f ms, free %7.1f ms\n", round, g,
                   (t1 - t0) * 1e3, (t1 - t0) * 1e3 / g, tt, tt2, (t3 - t2) * 1e3);
        }
    {
        hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
        uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
        for (int round = 0; round < 2; round++)
            for (size_t g : {8ull, 64ull}) {
                void* p = nullptr;
                double t0 = now(); hipError_t e = hipMallocAsync(&p, g * GB, 0); hipStreamSynchronize(0); double t1 = now();
                if (e != hipSuccess) { printf("hipMallocAsync %zu GB failed: %s\n", g, hipGetErrorString(e)); continue; }
                const double tt = touch(p, g * GB), tt2 = touch(p, g * GB);
                double t2 = now(); hipFreeAsync(p, 0); hipStreamSynchronize(0); double t3 = now();
                printf("round %d hipMallocAsync %3zu GB: alloc %8.1f ms (%5.1f ms/GB), first write %7.1f ms, second write %7.1f ms, free %7.1f ms\n", round, g,
                       (t1 - t0) * 1e3, (t1 - t0) * 1e3 / g, tt, tt2, (t3 - t2) * 1e3);
            }
    }
    return 0;
}
