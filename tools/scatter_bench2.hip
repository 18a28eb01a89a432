// What does an MI355X make of the store pattern of a radix pass, depending on WHO writes WHERE?  (tools/, not part of the library)
//
// 6.2 G 8-byte keys go to 512 streams (one per digit), RUN keys per stream and "tile", nothing else is computed.
//   global : tiles come from one counter and tile t writes keys [t RUN, (t + 1) RUN) of every stream -- neighbouring runs of a
//            stream are written by different workgroups (other CUs, other XCDs); this is the layout of a stable (ordered) pass;
//   private: workgroup w owns a contiguous piece of every stream and writes it run by run -- neighbouring runs come from the
//            same workgroup a tile apart; this is the layout an UNORDERED pass may choose (pass 0: no order to keep).
// Each with the streams' starts on 128-byte lines (every run = whole lines) or off them (every run cut by line borders), with
// plain or nontemporal stores, plus a plain device-to-device copy for scale (the measured peak).
// build: hipcc -O3 --offload-arch=gfx950 tools/scatter_bench2.hip -o gpurun_out/scatter_bench2 ; run: gpurun_out/scatter_bench2 [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int RUN, bool NT>
__global__ __launch_bounds__(512) void scatter_global(u64* out, u64 stride, u64 skew, u32 tiles, u32* counter) {
    __shared__ u32 tk;
    for (;;) {
        if (threadIdx.x == 0) tk = atomicAdd(counter, 1u);
        __syncthreads();
        const u32 t = tk;
        __syncthreads();
        if (t >= tiles) break;
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            const u32 s = (u32)i * 512 + threadIdx.x;
            const u32 d = s / RUN, r = s % RUN;
            u64* p = out + (u64)d * stride + (u64)((d * 7u) & 15u) * skew + (u64)t * RUN + r;
            const u64 v = ((u64)t << 20) | s;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

// workgroup w writes piece w of every stream: `iters` runs of RUN keys, starting at a line (skew 0) or (w + 3 d) mod 16 keys past one
template <int RUN, bool NT>
__global__ __launch_bounds__(512) void scatter_private(u64* out, u64 stride, u64 per, u64 skew, u32 iters) {
    const u32 w = blockIdx.x;
    for (u32 it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            const u32 s = (u32)i * 512 + threadIdx.x;
            const u32 d = s / RUN, r = s % RUN;
            u64* p = out + (u64)d * stride + (u64)w * per + (u64)((w + 3u * d) & 15u) * skew + (u64)it * RUN + r;
            const u64 v = ((u64)it << 20) | s;
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, u64 n16) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}

static std::vector<std::string> rows;
static float timed(hipEvent_t a, hipEvent_t b) { hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
static void report(const char* what, int run, bool aligned, bool nt, double gb, float ms) {
    char buf[512];
    printf("%-8s run %3d keys (%4d B) %-9s %-5s: %7.2f ms for %.1f GB = %.2f TB/s\n", what, run, run * 8, aligned ? "aligned" : "unaligned", nt ? "nt" : "plain", ms, gb, gb / ms);
    snprintf(buf, sizeof buf, "{\"layout\": \"%s\", \"run_keys\": %d, \"run_bytes\": %d, \"line_aligned\": %s, \"nontemporal\": %s, \"ms\": %.3f, \"GB\": %.2f, \"TBps\": %.3f}",
             what, run, run * 8, aligned ? "true" : "false", nt ? "true" : "false", ms, gb, gb / ms);
    rows.push_back(buf);
}

template <int RUN, bool NT>
static void run_global(u64* out, u64 n, bool aligned, u32* counter, hipEvent_t a, hipEvent_t b) {
    const u64 stride = (n / 512) & ~15ull;          // streams start on lines; `skew` moves them off
    const u32 tiles = (u32)((stride - 16) / RUN);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(counter, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL((scatter_global<RUN, NT>), dim3(512), dim3(512), 0, 0, out, stride, aligned ? 0ull : 1ull, tiles, counter);
        ms = timed(a, b);
    }
    report("global", RUN, aligned, NT, 8.0 * tiles * RUN * 512 / 1e9, ms);
}
template <int RUN, bool NT>
static void run_private(u64* out, u64 n, bool aligned, hipEvent_t a, hipEvent_t b) {
    const u64 stride = (n / 512) & ~15ull;
    const u64 per = (stride / 512) & ~15ull;
    const u32 iters = (u32)((per - 16) / RUN);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((scatter_private<RUN, NT>), dim3(512), dim3(512), 0, 0, out, stride, per, aligned ? 0ull : 1ull, iters);
        ms = timed(a, b);
    }
    report("private", RUN, aligned, NT, 8.0 * iters * RUN * 512 * 512 / 1e9, ms);
}

int main(int argc, char** argv) {
    const u64 n = 6221650873ull;
    u64* out; u32* counter;
    if (hipMalloc(&out, 8 * n) != hipSuccess || hipMalloc(&counter, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    run_global<16, false>(out, n, true, counter, a, b);
    run_global<16, false>(out, n, false, counter, a, b);
    run_global<16, true>(out, n, false, counter, a, b);
    run_private<16, false>(out, n, true, a, b);
    run_private<16, false>(out, n, false, a, b);
    run_private<16, true>(out, n, true, a, b);
    run_private<16, true>(out, n, false, a, b);
    run_private<8, false>(out, n, true, a, b);
    run_private<8, false>(out, n, false, a, b);
    run_private<32, false>(out, n, true, a, b);
    run_private<32, false>(out, n, false, a, b);
    run_global<32, false>(out, n, true, counter, a, b);
    run_global<32, false>(out, n, false, counter, a, b);
    // device-to-device copy: half of the buffer onto the other half, by a 16-byte-per-lane kernel and by hipMemcpyAsync
    const u64 half = (4 * n) & ~255ull;
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(copy16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)out, (uint4*)((char*)out + half), half / 16);
        ms = timed(a, b);
    }
    char buf[256];
    printf("copy16 kernel   : %7.2f ms, %.2f GB read + %.2f GB written = %.2f TB/s\n", ms, half / 1e9, half / 1e9, 2.0 * half / 1e9 / ms);
    snprintf(buf, sizeof buf, "{\"layout\": \"d2d copy, 16 B per lane kernel\", \"ms\": %.3f, \"GB_read\": %.2f, \"GB_written\": %.2f, \"TBps_read_plus_write\": %.3f}", ms, half / 1e9, half / 1e9, 2.0 * half / 1e9 / ms);
    rows.push_back(buf);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipMemcpyAsync((char*)out + half, out, half, hipMemcpyDeviceToDevice, 0);
        ms = timed(a, b);
    }
    printf("hipMemcpyAsync  : %7.2f ms = %.2f TB/s (read + write)\n", ms, 2.0 * half / 1e9 / ms);
    snprintf(buf, sizeof buf, "{\"layout\": \"d2d copy, hipMemcpyAsync\", \"ms\": %.3f, \"GB_read\": %.2f, \"GB_written\": %.2f, \"TBps_read_plus_write\": %.3f}", ms, half / 1e9, half / 1e9, 2.0 * half / 1e9 / ms);
    rows.push_back(buf);
    // read-only and write-only streams
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\"what\": \"tools/scatter_bench2.hip on MI355X: the store pattern of a radix pass (512 streams) and a device-to-device copy\", \"rows\": [\n");
            for (size_t i = 0; i < rows.size(); i++) fprintf(f, " %s%s\n", rows[i].c_str(), i + 1 < rows.size() ? "," : "");
            fprintf(f, "]}\n");
            fclose(f);
        }
    }
    return 0;
}
