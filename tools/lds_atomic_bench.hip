// What does an LDS atomic cost on MI355X?  (tools/, not part of the library)
// dedupe_kernel makes two LDS atomics per key (a compare-and-swap on the tag, an add on the count) into a 12 K-entry table, and
// stream_hist_kernel two adds per window into 512-bin histograms; both were read as "bound by the LDS atomic unit".  This measures
// the unit alone: one workgroup per CU, every wave makes ROUNDS x 8 operations of one kind, cycles by s_memtime around the loop.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench ; run: gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
typedef unsigned int u32;

enum Op { RD32, WR32, ADD32, ADDRTN32, CAS32, CAS_ADD, RD_THEN_ADD, CAS64, ADD64, NOPS };
static const char* op_name[NOPS] = {"ds_read_b32", "ds_write_b32", "ds_add_u32 (no return)", "ds_add_rtn_u32", "ds_cmpst_rtn_b32",
                                    "cmpst_rtn + add (dedupe insert)", "read, then add", "ds_cmpst_rtn_b64", "ds_add_u64 (no return)"};
enum Addr { RANDOM, LINEAR, SKEW512, SAME_BANK, NADDR };
static const char* addr_name[NADDR] = {"random in table", "lane-linear", "512 bins, skewed", "one bank"};

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int OP, int ADDR, int THREADS>
__global__ __launch_bounds__(THREADS) void bench(u32 entries, u32 rounds, u64* cycles, u32* sink) {
    extern __shared__ u32 tab[];
    for (u32 q = threadIdx.x; q < entries * 2; q += THREADS) tab[q] = 0xffffffffu;
    __syncthreads();
    u32 acc = 0;
    const u32 tid = threadIdx.x;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (u32 r = 0; r < rounds; r++) {
        u32 h[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32 x = mix((r * 8 + i) * 0x10001u + tid * 0x9E3779B1u + blockIdx.x);
            if (ADDR == RANDOM) h[i] = (u32)(((u64)x * entries) >> 32);
            else if (ADDR == LINEAR) h[i] = (tid + (r * 8 + i) * THREADS) % entries;
            else if (ADDR == SKEW512) { const u32 y = x & 511u; h[i] = (x >> 9 & 3u) ? y >> 2 : y; }          // 3/4 of the adds into the low 128 bins
            else h[i] = ((x % (entries / 32)) * 32);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == RD32) acc += tab[h[i]];
            else if (OP == WR32) tab[h[i]] = r;
            else if (OP == ADD32) atomicAdd(&tab[h[i]], 1u);
            else if (OP == ADDRTN32) acc += atomicAdd(&tab[h[i]], 1u);
            else if (OP == CAS32) acc += atomicCAS(&tab[h[i]], 0xffffffffu, h[i] * 3u);
            else if (OP == CAS_ADD) {
                const u32 old = atomicCAS(&tab[h[i]], 0xffffffffu, h[i] * 3u);
                atomicAdd(&tab[entries + h[i]], (old == 0xffffffffu || old == h[i] * 3u) ? 1u : 0u);
            } else if (OP == RD_THEN_ADD) {
                const u32 old = tab[h[i]];
                atomicAdd(&tab[entries + h[i]], old == 0xffffffffu ? 1u : 0u);
            } else if (OP == CAS64) {
                acc += (u32)atomicCAS(reinterpret_cast<unsigned long long*>(tab) + h[i], ~0ull, (unsigned long long)h[i]);
            } else if (OP == ADD64) {
                atomicAdd(reinterpret_cast<unsigned long long*>(tab) + h[i], 1ull);
            }
        }
    }
    __syncthreads();
    const u64 t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int OP, int ADDR, int THREADS>
static void run(u32 entries, int blocks_per_cu, u64* cycles, u32* sink) {
    const u32 rounds = 256;
    const int cus = 256;
    const size_t smem = (size_t)entries * 8;
    hipFuncSetAttribute((const void*)bench<OP, ADDR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((bench<OP, ADDR, THREADS>), dim3(cus * blocks_per_cu), dim3(THREADS), smem, 0, entries, rounds, cycles, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    std::vector<u64> c(cus * blocks_per_cu);
    hipMemcpy(c.data(), cycles, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[c.size() / 2];
    const double wave_instr = (double)rounds * 8 * (THREADS / 64) * blocks_per_cu;          // per CU
    const double lanes = wave_instr * 64;
    // s_memtime ticks at 100 MHz; the wall time of the launch gives the rate in seconds whatever the clock
    printf("%-34s %-18s %4d thr x %d/CU  table %5u: %8.3f ms  %7.1f ns per wave-op per CU  %6.2f lane-ops/ns per CU  (ticks %.0f)\n", op_name[OP], addr_name[ADDR],
           THREADS, blocks_per_cu, entries, ms, ms * 1e6 / wave_instr, lanes / (ms * 1e6), med);
}

int main() {
    u64* cycles; u32* sink;
    hipMalloc(&cycles, 8 * 1024); hipMalloc(&sink, 4);
    const u32 E = 12288;
    run<RD32, RANDOM, 1024>(E, 1, cycles, sink);
    run<WR32, RANDOM, 1024>(E, 1, cycles, sink);
    run<ADD32, RANDOM, 1024>(E, 1, cycles, sink);
    run<ADDRTN32, RANDOM, 1024>(E, 1, cycles, sink);
    run<CAS32, RANDOM, 1024>(E, 1, cycles, sink);
    run<CAS_ADD, RANDOM, 1024>(E, 1, cycles, sink);
    run<RD_THEN_ADD, RANDOM, 1024>(E, 1, cycles, sink);
    run<CAS64, RANDOM, 1024>(E / 2, 1, cycles, sink);
    run<ADD64, RANDOM, 1024>(E / 2, 1, cycles, sink);
    printf("\n");
    run<RD32, LINEAR, 1024>(E, 1, cycles, sink);
    run<WR32, LINEAR, 1024>(E, 1, cycles, sink);
    run<ADD32, LINEAR, 1024>(E, 1, cycles, sink);
    run<ADDRTN32, LINEAR, 1024>(E, 1, cycles, sink);
    run<CAS32, LINEAR, 1024>(E, 1, cycles, sink);
    run<CAS_ADD, LINEAR, 1024>(E, 1, cycles, sink);
    printf("\n");
    run<ADD32, SAME_BANK, 1024>(E, 1, cycles, sink);
    run<ADD32, SKEW512, 1024>(E, 1, cycles, sink);
    run<ADD32, SKEW512, 512>(4096, 4, cycles, sink);
    run<ADDRTN32, SKEW512, 512>(4096, 2, cycles, sink);
    printf("\n");
    run<CAS_ADD, RANDOM, 512>(8192, 2, cycles, sink);
    run<CAS_ADD, RANDOM, 512>(E, 1, cycles, sink);
    run<CAS_ADD, RANDOM, 256>(4096, 4, cycles, sink);
    run<ADD32, RANDOM, 256>(4096, 4, cycles, sink);
    run<ADD32, RANDOM, 256>(4096, 8, cycles, sink);
    return 0;
}
