// partition.hip -- the device side of the multi-GPU exchange and the synthetic
// k-mer sets of BASELINE configs 3 and 4.
//
//   * hash-range owner: owner(x) = floor(murmer(x, seed) * world / 2^64)  (basics.murmer,
//     zotmer/library/basics.py:191-229) -- equal k-mers meet on one GPU whatever their value, so skewed
//     genomes (poly-A, low complexity) spread evenly.  A sorted (k-mer, count) table is split into `world`
//     pieces, each still sorted: a STABLE multi-way partition in three launches -- per-tile owner counts,
//     one 64-bit scan of the [owner][tile] table, scatter.  (A value-range owner needs none of this: a sorted
//     table is already partitioned, the cut points are binary searches -- zk_lower_bound.)
//   * order-free checksums over 32- or 64-bit counts (`zot merge` works on 64-bit counts).
//   * counter-based generators for sorted k-mer sets (zotmer_amd/synth.py is the specification).
//
// Algorithmic bytes of the partition: keys read twice + counts read once + both written once.
#include <vector>

#include "internal.hpp"

namespace zk {

constexpr int HP_BLOCK = 256;
constexpr int HP_ITEMS = 8;
constexpr int HP_TILE = HP_BLOCK * HP_ITEMS;
constexpr int HP_NW = HP_BLOCK / 64;
constexpr int HP_MAXW = 32;

__device__ __forceinline__ u32 hash_owner(u64 x, u64 seed, u32 world) { return (u32)__umul64hi(murmer(x, seed), (u64)world); }

// SCATTER == false: table[o * tiles + tile] = number of the tile's elements owned by o.
// SCATTER == true : table holds the inclusive scan of those counts; element -> its slot.  Order inside a piece is
//                   tile order, then wave, row, lane = input order, so every piece stays sorted.
template <typename CT, bool SCATTER>
__global__ __launch_bounds__(HP_BLOCK) void hash_part_kernel(const u64* __restrict__ keys, const CT* __restrict__ cnts, u64 n, u32 world,
                                                             u64 seed, u32 tiles, u64* __restrict__ table, u64* __restrict__ ok,
                                                             CT* __restrict__ oc) {
    __shared__ u32 wrun_s[HP_NW][HP_MAXW];
    __shared__ u64 goff[HP_MAXW];
    volatile u32(*wrun)[HP_MAXW] = wrun_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile = blockIdx.x;
    if (tid < HP_NW * HP_MAXW) ((u32*)wrun_s)[tid] = 0;
    __syncthreads();
    const u64 base = (u64)tile * HP_TILE + (u64)wave * (64 * HP_ITEMS);
    u64 k[HP_ITEMS];
    u32 own[HP_ITEMS], rank[HP_ITEMS];
#pragma unroll
    for (int i = 0; i < HP_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        k[i] = (idx < n) ? keys[idx] : 0ull;
        own[i] = (idx < n) ? hash_owner(k[i], seed, world) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int i = 0; i < HP_ITEMS; i++) {
        u64 rem = __ballot(own[i] != 0xFFFFFFFFu);
        rank[i] = 0;
        while (rem) {                                              // one round per distinct owner in the row (<= world)
            const int leader = __builtin_ctzll(rem);
            const u32 d = (u32)__shfl((int)own[i], leader, 64);
            const u64 m = __ballot(own[i] == d);
            if (own[i] == d) rank[i] = wrun[wave][d] + popc_below(m);
            if (lane == leader) wrun[wave][d] = wrun[wave][d] + (u32)__popcll(m);
            rem &= ~m;
        }
    }
    __syncthreads();
    if (!SCATTER) {
        if (tid < (int)world) {
            u32 t = 0;
#pragma unroll
            for (int w = 0; w < HP_NW; w++) t += wrun[w][tid];
            table[(u64)tid * tiles + tile] = t;
        }
        return;
    }
    if (tid < (int)world) {
        const u64 flat = (u64)tid * tiles + tile;
        goff[tid] = flat ? table[flat - 1] : 0ull;
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < HP_NW; w++) { const u32 t = wrun[w][tid]; wrun[w][tid] = run; run += t; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < HP_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        if (idx < n) {
            const u64 pos = goff[own[i]] + wrun[wave][own[i]] + rank[i];
            ok[pos] = k[i];
            if (cnts) oc[pos] = cnts[idx];
        }
    }
}

// offsets[o] = first slot of owner o's piece (o = world: n)
__global__ void hash_offsets_kernel(const u64* __restrict__ table, u32 tiles, u32 world, u64 n, u64* __restrict__ out) {
    const u32 o = threadIdx.x;
    if (o > world) return;
    out[o] = (o == world) ? n : (o == 0 ? 0ull : table[(u64)o * tiles - 1]);
}

template <typename CT>
static int hash_partition_t(zk_ctx* c, const u64* keys, const CT* cnts, uint64_t n, int world, u64 seed, u64* ok, CT* oc,
                            uint64_t* offsets) {
    for (int o = 0; o <= world; o++) offsets[o] = 0;
    if (n == 0) return ZK_OK;
    const u32 tiles = (u32)div_up(n, HP_TILE);
    const uint64_t words = (uint64_t)world * tiles;
    u64 *table, *d_off;
    ZK_TRY(arena_require(c, 8 * words + (1 << 20), 8 * words + (1 << 20)));
    ZK_TRY(arena_alloc(c, 8 * words, (void**)&table));
    ZK_TRY(arena_alloc(c, 8 * (HP_MAXW + 1), (void**)&d_off));
    prof_begin(c, ZK_PROF_SELECT, (16 + 2 * (cnts ? sizeof(CT) : 0) + 8) * n);
    hipLaunchKernelGGL((hash_part_kernel<CT, false>), dim3(tiles), dim3(HP_BLOCK), 0, c->stream, keys, cnts, (u64)n, (u32)world, seed, tiles,
                       table, ok, oc);
    ZK_HIP(c, hipGetLastError());
    ZK_TRY(scan64_inclusive(c, table, words));
    hipLaunchKernelGGL((hash_part_kernel<CT, true>), dim3(tiles), dim3(HP_BLOCK), 0, c->stream, keys, cnts, (u64)n, (u32)world, seed, tiles,
                       table, ok, oc);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(hash_offsets_kernel, dim3(1), dim3(64), 0, c->stream, table, tiles, (u32)world, (u64)n, d_off);
    ZK_HIP(c, hipGetLastError());
    std::vector<u64> h(world + 1);
    ZK_HIP(c, hipMemcpyAsync(h.data(), d_off, 8 * (world + 1), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (int o = 0; o <= world; o++) offsets[o] = h[o];
    return check_device_error(c);
}

int hash_partition(zk_ctx* c, const u64* keys, const void* cnts, int count_bits, uint64_t n, int world, u64 seed, u64* ok, void* oc,
                   uint64_t* offsets) {
    if (world < 1 || world > HP_MAXW) return fail(c, ZK_EINVAL, "hash partition: world must be in 1..%d (got %d)", HP_MAXW, world);
    if (count_bits == 32) return hash_partition_t<u32>(c, keys, (const u32*)cnts, n, world, seed, ok, (u32*)oc, offsets);
    return hash_partition_t<u64>(c, keys, (const u64*)cnts, n, world, seed, ok, (u64*)oc, offsets);
}

// ---- order-free checksums over any count width ---------------------------------------------------------
template <typename CT>
__global__ void checksum_any_kernel(const u64* __restrict__ k, const CT* __restrict__ cn, u64 n, u64* sums) {
    u64 s0 = 0, s1 = 0, s2 = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 w = cn ? (u64)cn[i] : 1ull;
        s0 += w; s1 += k[i] * w; s2 += murmer(k[i], 0) * w;
    }
    s0 = wave_sum_u64(s0); s1 = wave_sum_u64(s1); s2 = wave_sum_u64(s2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], s2); }
}

int checksum_any(zk_ctx* c, const u64* keys, const void* cnts, int count_bits, uint64_t n, uint64_t sums[3]) {
    u64* d = c->d_scalars + 12;
    ZK_HIP(c, hipMemsetAsync(d, 0, 3 * sizeof(u64), c->stream));
    if (n) {
        u64 g = div_up(n, 256 * 8), mx = (u64)c->num_cus * 16;
        const u32 grid = (u32)(g < mx ? g : mx);
        if (count_bits == 32)
            hipLaunchKernelGGL((checksum_any_kernel<u32>), dim3(grid), dim3(256), 0, c->stream, keys, (const u32*)cnts, (u64)n, d);
        else
            hipLaunchKernelGGL((checksum_any_kernel<u64>), dim3(grid), dim3(256), 0, c->stream, keys, (const u64*)cnts, (u64)n, d);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 12, d, 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) sums[i] = c->h_scalars[12 + i];
    return ZK_OK;
}

// ---- synthetic k-mer sets (zotmer_amd/synth.py: set_keys / set_counts) -----------------------------------
__device__ __forceinline__ u64 smix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// out[i] = rnd(seed, 7, (mul * (first + i) + add) % mod) & mask: element (first + i) of an affine walk through a pool
// of `mod` keys -- distinct pool indices while first + i < mod and gcd(mul, mod) = 1, i.e. a draw without replacement.
__global__ void synth_keys_kernel(u64 seed, u64 first, u64 count, u64 mul, u64 add, u64 mod, u64 mask, u64* __restrict__ out) {
    const u64 s7 = smix64(seed + 7);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (u64)gridDim.x * blockDim.x) {
        const u64 j = (mul * (first + i) + add) % mod;
        out[i] = smix64(s7 + j) & mask;
    }
}

// geometric counts, mean 8, in integer arithmetic: successive 3-bit groups of rnd(seed, 8, key) are trials that stop
// with probability 1/8; a word whose 21 groups all fail is re-mixed (at most 8 words).
__global__ void synth_counts_kernel(u64 seed, const u64* __restrict__ keys, u64 n, u64* __restrict__ out) {
    const u64 s8 = smix64(seed + 8);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        u64 h = smix64(s8 + keys[i]);
        u64 cnt = 1;
        bool done = false;
        for (int round = 0; round < 8 && !done; round++) {
            for (int g = 0; g < 21; g++) {
                if (((h >> (3 * g)) & 7ull) == 0) { done = true; break; }
                cnt++;
            }
            h = smix64(h);
        }
        out[i] = cnt;
    }
}

// basics.can (zotmer/library/basics.py:231-250): of x and rc(x), the one with the smaller murmer hash (seed 17), x on a tie
__global__ void can_kernel(const u64* __restrict__ in, u64 n, int K, u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 x = in[i], xb = revcomp(K, x);
        out[i] = (murmer(x, 17) <= murmer(xb, 17)) ? x : xb;
    }
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_can(zk_ctx* c, int K, const uint64_t* d_kmers, uint64_t n, uint64_t* d_out) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (K < 1 || K > 32 || (n && (!d_kmers || !d_out))) return fail(c, ZK_EINVAL, "zk_can: bad argument");
    if (n == 0) return ZK_OK;
    u64 g = div_up(n, 256 * 8), mx = (u64)c->num_cus * 16;
    hipLaunchKernelGGL(can_kernel, dim3((u32)(g < mx ? g : mx)), dim3(256), 0, c->stream, (const u64*)d_kmers, (u64)n, K, (u64*)d_out);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int zk_hash_partition(zk_ctx* c, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n, int world, uint64_t seed,
                      uint64_t* d_ok, void* d_oc, uint64_t* offsets) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (!offsets || (count_bits != 32 && count_bits != 64) || (n && (!d_kmers || !d_ok)) || (d_counts && !d_oc))
        return fail(c, ZK_EINVAL, "zk_hash_partition: bad argument");
    arena_reset(c);
    return hash_partition(c, (const u64*)d_kmers, d_counts, count_bits, n, world, seed, (u64*)d_ok, d_oc, offsets);
}

// first index whose key is not above the one before it (n: strictly ascending)
__global__ void first_descent_kernel(const u64* __restrict__ k, u64 n, u64* out) {
    u64 best = ~0ull;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (u64)gridDim.x * blockDim.x)
        if (k[i] <= k[i - 1] && i < best) best = i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(best, o, 64); best = t < best ? t : best; }
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin((unsigned long long*)out, (unsigned long long)best);
}

int zk_first_descent(zk_ctx* c, const uint64_t* d_kmers, uint64_t n, uint64_t* first_bad) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (!first_bad || (n && !d_kmers)) return fail(c, ZK_EINVAL, "zk_first_descent: bad argument");
    u64* d = c->d_scalars + 12;
    ZK_HIP(c, hipMemsetAsync(d, 0xff, sizeof(u64), c->stream));
    if (n > 1) {
        const u64 g = (n + 255) / 256, mx = (u64)c->num_cus * 16;
        hipLaunchKernelGGL(first_descent_kernel, dim3((u32)(g < mx ? g : mx)), dim3(256), 0, c->stream, (const u64*)d_kmers, (u64)n, d);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 12, d, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *first_bad = c->h_scalars[12] == ~0ull ? n : c->h_scalars[12];
    return ZK_OK;
}

int zk_checksum_counts(zk_ctx* c, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n, uint64_t sums[3]) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (!sums || (count_bits != 32 && count_bits != 64)) return fail(c, ZK_EINVAL, "zk_checksum_counts: bad argument");
    return checksum_any(c, (const u64*)d_kmers, d_counts, count_bits, n, sums);
}

int zk_synth_keys(zk_ctx* c, uint64_t seed, uint64_t first, uint64_t count, int key_bits, uint64_t mul, uint64_t add, uint64_t mod,
                  uint64_t* d_out) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (key_bits < 1 || key_bits > 64 || mod == 0 || mul == 0) return fail(c, ZK_EINVAL, "zk_synth_keys: bad argument");
    // mul * (first + count) + add must not wrap
    const unsigned __int128 top = (unsigned __int128)mul * (first + count) + add;
    if (top >> 64) return fail(c, ZK_EINVAL, "zk_synth_keys: mul * (first + count) + add overflows 64 bits");
    if (count == 0) return ZK_OK;
    const u64 mask = key_bits == 64 ? ~0ull : ((1ull << key_bits) - 1);
    u64 g = div_up(count, 256 * 8), mx = (u64)c->num_cus * 16;
    hipLaunchKernelGGL(synth_keys_kernel, dim3((u32)(g < mx ? g : mx)), dim3(256), 0, c->stream, (u64)seed, (u64)first, (u64)count,
                       (u64)mul, (u64)add, (u64)mod, mask, (u64*)d_out);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int zk_synth_counts(zk_ctx* c, uint64_t seed, const uint64_t* d_keys, uint64_t n, uint64_t* d_counts) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (n == 0) return ZK_OK;
    u64 g = div_up(n, 256 * 8), mx = (u64)c->num_cus * 16;
    hipLaunchKernelGGL(synth_counts_kernel, dim3((u32)(g < mx ? g : mx)), dim3(256), 0, c->stream, (u64)seed, (const u64*)d_keys, (u64)n,
                       (u64*)d_counts);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

}  // extern "C"
