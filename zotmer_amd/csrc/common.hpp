// common.hpp -- shared device helpers for libzotk (gfx950 / CDNA4, wave64 only).
//
// Everything here is integer/bit work bounded by HBM bandwidth: no MFMA anywhere.
// Conventions:
//   * a wavefront is 64 lanes; 64 is hard-coded (cdna_hip_programming.md section 1);
//   * inter-workgroup hand-offs use ONE naturally aligned 64-bit word that carries both the
//     flag and the payload, written and polled with relaxed agent-scope atomics, so no
//     separate release/acquire is needed (MI355X_MICROARCH.md, "Valid forms": 8-byte agent
//     atomics on both sides);
//   * every spin is bounded and reports through the context's error word instead of hanging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

#define ZK_WAVE 64

// bits of the device error word (zk_ctx::d_err)
#define ZK_DERR_SPIN_TIMEOUT 1u   // a look-back spin gave up (would otherwise hang)
#define ZK_DERR_COUNT_OVERFLOW 2u // a 32-bit count wrapped
#define ZK_DERR_CAPACITY 4u       // an output did not fit the caller's buffer
#define ZK_DERR_RANGE 8u          // codec64: a value (or k-mer delta) >= 2^60 has no code
#define ZK_DERR_BAD_TAG 16u       // codec64: a word carries a tag the format does not define
#define ZK_DERR_MISMATCH 32u      // stream_pass.hip: the pass did not write what the histogram had counted for it
#define ZK_DERR_SHARED_KEY 64u    // setops.hip: two lists merged as disjoint share a key

namespace zk {

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// the XCD this wave runs on: HW_REG_XCC_ID (register 20), bits [3:0]
__device__ __forceinline__ u32 xcc_id() { return (u32)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u; }

// number of set bits of `m` strictly below the calling lane
__device__ __forceinline__ u32 popc_below(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u32 t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        u64 t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ u64 ld_agent(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------
// Decoupled look-back status word:  [63:62] state  [61:57] epoch  [56:0] value
// A word whose epoch differs from the launch's epoch reads as EMPTY, so the status array is
// cleared once per 31 launches instead of once per launch (epochs run 1..31; 0 = cleared).
// ---------------------------------------------------------------------------------------
#define ZK_ST_PARTIAL 1ull
#define ZK_ST_INCLUSIVE 2ull
#define ZK_ST_VALUE_MASK ((1ull << 57) - 1)
#define ZK_SPIN_LIMIT (1 << 22)

__device__ __forceinline__ u64 st_pack(u64 state, u32 epoch, u64 value) {
    return (state << 62) | ((u64)epoch << 57) | (value & ZK_ST_VALUE_MASK);
}
__device__ __forceinline__ u64 st_state(u64 w, u32 epoch) {
    return (((w >> 57) & 31u) == epoch) ? (w >> 62) : 0ull;
}

// Dynamic tile id: tiles are numbered in the order their workgroups start, so every tile a
// look-back waits for is already running or done -- forward progress does not depend on the
// dispatch order (which HIP does not define).
__device__ __forceinline__ u32 take_ticket(u32* counter, u32* lds_slot) {
    if (threadIdx.x == 0) *lds_slot = atomicAdd(counter, 1u);
    __syncthreads();
    return (u32)__builtin_amdgcn_readfirstlane((int)*lds_slot);   // the same in every lane: keep it in an SGPR
}

// One value per tile.  Called by all 64 lanes of ONE wave of the workgroup; returns the sum of
// `total` over all tiles with a smaller id (same value in every lane) and publishes this
// tile's inclusive prefix.
__device__ __forceinline__ u64 lookback_exclusive(u64* status, u32 tile, u64 total, u32 epoch, u32* err) {
    const int l = lane_id();
    if (tile == 0) {
        if (l == 0) st_agent(&status[0], st_pack(ZK_ST_INCLUSIVE, epoch, total));
        return 0;
    }
    if (l == 0) st_agent(&status[tile], st_pack(ZK_ST_PARTIAL, epoch, total));
    u64 excl = 0;
    long long idx = (long long)tile - 1 - l;   // lane 0 looks at the nearest predecessor
    int spins = 0;
    while (true) {
        u64 w = (idx >= 0) ? ld_agent(&status[idx]) : st_pack(ZK_ST_INCLUSIVE, epoch, 0);
        u64 s = st_state(w, epoch);
        // lanes past the first INCLUSIVE one do not matter; wait only for the ones before it
        u64 incl = __ballot(s == ZK_ST_INCLUSIVE);
        u64 empty = __ballot(s == 0);
        u64 need = incl ? ((incl & (0ull - incl)) - 1) | (incl & (0ull - incl)) : ~0ull;  // lanes <= first inclusive
        if (empty & need) {
            if (++spins > ZK_SPIN_LIMIT) {
                if (l == 0) atomicOr(err, ZK_DERR_SPIN_TIMEOUT | (512u << 8));
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        u64 take = ((need >> l) & 1ull) ? (w & ZK_ST_VALUE_MASK) : 0ull;
        excl += wave_sum_u64(take);
        if (incl) break;
        idx -= 64;
    }
    if (l == 0) st_agent(&status[tile], st_pack(ZK_ST_INCLUSIVE, epoch, excl + total));
    return excl;
}

// The decoupled look-back in two steps, so that work can be put between them: a tile publishes its count as soon as it has
// it (one thread), and finds the sum over the tiles before it later (one wave; publishes the inclusive prefix)
__device__ __forceinline__ void lookback_publish(u64* status, u32 tile, u64 total, u32 epoch) {
    st_agent(&status[tile], st_pack(tile == 0 ? ZK_ST_INCLUSIVE : ZK_ST_PARTIAL, epoch, total));
}
__device__ __forceinline__ u64 lookback_resolve(u64* status, u32 tile, u64 total, u32 epoch, u32* err) {
    const int l = lane_id();
    if (tile == 0) return 0;
    u64 excl = 0;
    long long idx = (long long)tile - 1 - l;   // lane 0 looks at the nearest predecessor
    int spins = 0;
    while (true) {
        const u64 w = (idx >= 0) ? ld_agent(&status[idx]) : st_pack(ZK_ST_INCLUSIVE, epoch, 0);
        const u64 s = st_state(w, epoch);
        const u64 incl = __ballot(s == ZK_ST_INCLUSIVE);
        const u64 empty = __ballot(s == 0);
        const u64 need = incl ? ((incl & (0ull - incl)) - 1) | (incl & (0ull - incl)) : ~0ull;  // lanes <= first inclusive
        if (empty & need) {
            if (++spins > ZK_SPIN_LIMIT) {
                if (l == 0) atomicOr(err, ZK_DERR_SPIN_TIMEOUT | (0x40u << 8));
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const u64 take = ((need >> l) & 1ull) ? (w & ZK_ST_VALUE_MASK) : 0ull;
        excl += wave_sum_u64(take);
        if (incl) break;
        idx -= 64;
    }
    if (l == 0) st_agent(&status[tile], st_pack(ZK_ST_INCLUSIVE, epoch, excl + total));
    return excl;
}

// Sum of v over the workgroup, valid in thread 0 (scratch: one u64 per wave).  Ends with a barrier.
__device__ __forceinline__ u64 block_sum_u64(u64 v, u64* scratch) {
    v = wave_sum_u64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    u64 t = 0;
    if (threadIdx.x == 0) for (int w = 0; w < nw; w++) t += scratch[w];
    __syncthreads();
    return t;
}

// reverse the 32 bit-pairs of a word (bits.rev, zotmer/library/bits.py:22-31)
__device__ __forceinline__ u64 rev_pairs(u64 x) {
    u64 y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}
// basics.rc (zotmer/library/basics.py:115-121)
__device__ __forceinline__ u64 revcomp(int K, u64 x) { return rev_pairs(~x) >> (64 - 2 * K); }

// basics.murmer (zotmer/library/basics.py:191-229)
__device__ __forceinline__ u64 murmer(u64 x, u64 s) {
    u64 k = x * 0x87c37b91114253d5ull;
    k = (k << 31) | (k >> 33);
    k *= 0x4cf5ad432745937full;
    u64 h = s ^ k;
    h = (h << 27) | (h >> 37);
    h = h * 5 + 0x52dce729ull;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

}  // namespace zk
