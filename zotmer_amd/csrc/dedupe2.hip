// dedupe2.hip -- K4, the block dedupe with TWO workgroups per CU (32-bit tags only).
//
// Replaces the run-length half of kmerize.merge (zotmer/commands/kmerize.py:41-132) exactly as radix_sort.hip::dedupe_kernel does;
// same inputs, same words out.  What is different is how a CU is used.  dedupe_kernel's table (12 K entries of 4 + 4 bytes, 142 KB
// with its lists) leaves room for one workgroup per CU, and a block's phases -- clear, insert, drain, group, rank, write -- run one
// after the other with a barrier between them: tools/lds_atomic_bench.hip measures the two LDS atomics of an insert at 10.3 ns per
// 64 keys per CU (8 K cycles for a block of 23.7 K keys), the kernel needed 45 K cycles per block.  Nothing in it is short of a unit;
// it is short of something else to run while a phase waits.  So: a table that fits twice --
//   * counts of 16 bits, two to a word (a block of fewer than 65 536 keys cannot carry from one into the other; larger blocks are
//     declined up front and counted by dedupe_kernel afterwards, as are the blocks whose table fills up);
//   * side lists of 128 entries a wave, drained 64 at a time as soon as 64 are there (full wavefronts, no worst-case tile to hold);
// -- 11 K entries in 76 KB, and while one workgroup of the CU sorts and writes its block the other one inserts.
#include "dedupe.hpp"
#include <type_traits>

namespace zk {

// BATCH: the compare-and-swaps a thread has in flight at a time (their answers are looked at together); a wave's side list must
// take what one batch can add at worst on top of the 63 entries a drain may leave.  ITEMS: tags a thread takes per tile; the next
// tile is asked for before the current one is inserted, so a workgroup has 4 * TILE bytes on their way from memory.
template <int BLOCK_, int BATCH_, int ITEMS_>
struct Dedupe2Smem {
    static constexpr int BLOCK = BLOCK_, BATCH = BATCH_, ITEMS = ITEMS_, TILE = BLOCK * ITEMS, NW = BLOCK / 64, ALL = 11264, SPT = ALL / BLOCK,
                         NB = 1024, SIDE = 64 + 64 * BATCH;
    static_assert(ALL % BLOCK == 0 && ALL % 8 == 0 && ITEMS % BATCH == 0 && ITEMS % 4 == 0, "whole rounds, whole quads, whole batches");
    alignas(16) u32 keys[ALL];            // tags (after the count: the entries again, grouped by their top ten bits)
    alignas(16) u32 cnt[ALL / 2];         // 16-bit counts: entry h in half (h & 1) of word h >> 1
    union {
        u32 side[NW][SIDE];               // while the keys are inserted
        struct { u32 bc[NB]; u32 bbase[NB + 1]; } g;          // afterwards: entries per group (the tag's top ten bits), and before it
    };
    u32 ticket;
};

// what a workgroup carries from one block to the next: the block it is about to count, with the first tile of its tags already
// asked for -- the ticket, the bounds and those tags travel while the previous block is being sorted and written
template <int ITEMS>
struct Dedupe2Next {
    u32 chunk;
    u64 lo;          // where the block's keys start ...
    u32 len;         // ... and how many they are (a block this kernel counts has fewer than 65 536)
    u32 tag[ITEMS];
};

// The tags [pos, pos + TILE) of the block that starts at `lo` (the same in every lane), as far as the block goes (rem = len - pos > 0).
// A whole tile: every thread takes ITEMS tags with 16-byte loads (TAGIN; a block starts wherever it starts: 4-byte aligned, no more) or
// ITEMS keys a stride apart.  The cut last tile: element i * BLOCK + tid each, the ones beyond the block not loaded.  Which thread
// takes which key is the table's business alone.
template <int BLOCK, int ITEMS, bool TAGIN>
__device__ __forceinline__ void dedupe2_load(const DedupeArgs& a, u64 lo, u32 pos, u32 rem, u32 tmask, u32 (&t)[ITEMS]) {
    constexpr u32 TILE = BLOCK * ITEMS;
    u32 tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // (opaque: or every index derived from it is computed once, as a 64-bit pair, and kept for the whole kernel)
    if constexpr (TAGIN) {
        const u32* tp = a.tin + lo + pos;
        if (rem >= TILE) {
            struct __attribute__((packed, aligned(4))) Tag4 { u32 a, b, c, d; };
            static_assert(ITEMS % 4 == 0, "whole quads");
#pragma unroll
            for (int i = 0; i < ITEMS / 4; i++) {
                const Tag4 q = *reinterpret_cast<const Tag4*>(tp + (u32)i * (4 * BLOCK) + 4 * tid);
                t[4 * i] = q.a; t[4 * i + 1] = q.b; t[4 * i + 2] = q.c; t[4 * i + 3] = q.d;
            }
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u32 g = (u32)i * BLOCK + tid;
                t[i] = g < rem ? tp[g] : 0u;
            }
        }
    } else {
        const u64* kp = a.kin + lo + pos;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 g = (u32)i * BLOCK + tid;
            t[i] = g < rem ? (u32)kp[g] & tmask : 0u;
        }
    }
}

template <int BLOCK, int BATCH, int ITEMS_, bool TAGIN>
__device__ __forceinline__ void dedupe2_block(const DedupeArgs& a, Dedupe2Smem<BLOCK, BATCH, ITEMS_>& sm, Dedupe2Next<ITEMS_>& st, u32 (&ph)[8], u32& tlast) {
    using S = Dedupe2Smem<BLOCK, BATCH, ITEMS_>;
    constexpr int ITEMS = S::ITEMS, TILE = S::TILE, ALL = S::ALL, SPT = S::SPT, NB = S::NB;
    constexpr u32 EMPTY = ~0u;            // no entry; the all-ones tag has the last entry to itself (see `home`)
    constexpr u32 HS = ALL - 1;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // (opaque per block: nothing derived from it is kept from one block to the next)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 chunk = st.chunk;
    const u64 lo = st.lo;
    const u32 len = st.len;
    const u32 maxc = (1u << a.pack) - 1u;
    const u32 tmask = a.tag_bits >= 32 ? ~0u : (1u << a.tag_bits) - 1u;
    auto next_block = [&](u32 t) {          // t: the ticket drawn at this block's start (read after a barrier)
        st.chunk = t; st.lo = 0; st.len = 0;
        if (t < a.chunks) {
            const u64 nlo = a.cuts[t], nhi = a.cuts[t + 1];
            st.lo = nlo;
            st.len = nhi - nlo < (u64)a.limit ? (u32)(nhi - nlo) : ~0u;          // ~0: too large for this kernel, nothing is loaded
        }
        if (st.len - 1u < ~0u - 1u) dedupe2_load<BLOCK, ITEMS, TAGIN>(a, st.lo, 0, st.len, tmask, st.tag);
    };
    if (tid == 0) sm.ticket = atomicAdd(a.counter, 1u);          // the block after this one: read after the next barrier
    if (len == 0 || len == ~0u) {
        // nothing to count -- or more keys than a 16-bit count is safe for: dedupe_kernel's business
        if (tid == 0) {
            if (len) a.retry[atomicAdd(a.n_retry, 1u)] = chunk;
            a.nwords[chunk] = 0;
        }
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        __syncthreads();
        next_block((u32)__builtin_amdgcn_readfirstlane((int)sm.ticket));
        return;
    }
    for (int q = tid; q < ALL / 4; q += BLOCK) reinterpret_cast<uint4*>(sm.keys)[q] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int q = tid; q < ALL / 8; q += BLOCK) reinterpret_cast<uint4*>(sm.cnt)[q] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    DD_PHASE(0);          // table cleared
    const u32 nchunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    u32 bad = 0;
    u32 nside = 0;          // entries in this wave's side list (the same in every lane)
    auto home = [&](u32 e) -> u32 {
        // the all-ones tag (= the empty marker) has the last entry to itself: there the swap of "empty" for "empty" succeeds and
        // leaves the word as it is; no other key is ever sent there
        u32 hh = __umulhi(e * 0x9E3779B1u, HS);
        asm("" : "+v"(hh));          // (opaque: left to itself the compiler branches around the two multiplies for the one tag that does not need them)
        return e == EMPTY ? HS : hh;          // (24-bit multiplies instead -- fold, multiply, scale: six full-rate operations -- no faster: 12.4 ms)
    };
    auto count = [&](u32 h, u32 inc) { atomicAdd(&sm.cnt[h >> 1], inc << ((h & 1u) << 4)); };
    // up to 64 entries of the wave's side list into the table by linear probing
    auto drain = [&](u32 from, u32 n) {
        if ((u32)lane < n) {
            const u32 e = sm.side[wave][from + lane];
            u32 h = home(e) + 1;          // its home entry is taken: that is why it is here
            h = h >= HS ? 0u : h;
            int p = 0;
            for (; p < ALL; p++) {
                const u32 old = atomicCAS(&sm.keys[h], EMPTY, e);
                if (old == EMPTY || old == e) break;
                h = h + 1 == HS ? 0u : h + 1;
            }
            if (p < ALL) count(h, 1u); else bad = 1;
        }
    };
    // The common case has no loop and no branch: one compare-and-swap at the tag's home entry, the count added as 1 or 0 (adding 0
    // to another key's entry harms nobody); a key that finds another key at home goes to the wave's side list (its place from a
    // ballot, no atomic).  BATCH swaps are issued before the first answer is looked at.  (A plain read first and the swap only for
    // the lanes that see "empty": 13.5 against 13.3 ms.)
    auto insert = [&](auto whole, const u32 (&e)[BATCH], const bool (&valid)[BATCH]) {          // whole: every key of the batch is one (a whole tile)
        constexpr bool WHOLE = decltype(whole)::value;
        u32 h[BATCH], old[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) h[j] = home(e[j]);
#pragma unroll
        for (int j = 0; j < BATCH; j++) old[j] = (WHOLE || valid[j]) ? atomicCAS(&sm.keys[h[j]], EMPTY, e[j]) : e[j];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const bool ok = old[j] == EMPTY || old[j] == e[j];
            count(h[j], (ok && (WHOLE || valid[j])) ? 1u : 0u);
            const u64 m = __ballot(!ok);
            if (!ok) sm.side[wave][nside + popc_below(m)] = e[j];
            nside += (u32)__popcll(m);
        }
        while (nside >= 64) { nside -= 64; drain(nside, 64); }
    };
    u32 tag[ITEMS], nt[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) tag[i] = st.tag[i];          // the first tile was asked for during the previous block
    for (u32 pos = 0; pos < len; pos += TILE) {
        const u32 rem = len - pos;
        if (rem > (u32)TILE) dedupe2_load<BLOCK, ITEMS, TAGIN>(a, lo, pos + TILE, rem - TILE, tmask, nt);
#pragma unroll
        for (int i0 = 0; i0 < ITEMS; i0 += BATCH) {
            u32 e[BATCH];
            bool valid[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; j++) { e[j] = tag[i0 + j]; valid[j] = (u32)(i0 + j) * BLOCK + tid < rem; }
            if (rem >= (u32)TILE) insert(std::true_type(), e, valid); else insert(std::false_type(), e, valid);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) tag[i] = nt[i];
    }
    drain(0, nside);
    DD_PHASE(1);          // keys inserted
    const int any_bad = __syncthreads_or((int)bad);
    DD_PHASE(2);          // ... every wave done
    if (any_bad) {
        // the table filled up (a block with more distinct keys than it holds): dedupe_kernel's larger table gets a try
        if (tid == 0) {
            a.retry[atomicAdd(a.n_retry, 1u)] = chunk;
            a.nwords[chunk] = 0;
        }
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        next_block(nchunk);
        return;
    }
    // ---- the block's entries, sorted: a counting sort on the tag's top ten bits, then ranks inside each group of ~3 -----------
    // thread t takes the entries t, t + BLOCK, ... into registers with their places in their groups (a returning add); the table's
    // memory then takes them back grouped.  (256 groups of a dozen entries: the rank loop runs as long as the largest group of a
    // wavefront, one LDS round trip per turn -- 16 K of a block's 58 K cycles.)
    for (int q = tid; q < NB; q += BLOCK) sm.g.bc[q] = 0;          // the side lists are done with: every wave has passed the barrier above
    u32 et[SPT], ec[SPT];          // ec: count | place in the group << 16
    const int gb = a.tag_bits < 10 ? a.tag_bits : 10, bsh = a.tag_bits - gb;
    const u16* cnt16 = reinterpret_cast<const u16*>(sm.cnt);
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        et[j] = sm.keys[tid + j * BLOCK];
        ec[j] = cnt16[tid + j * BLOCK];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        if (ec[j]) ec[j] |= atomicAdd(&sm.g.bc[(et[j] >> bsh) & (NB - 1)], 1u) << 16;
        if (j % 6 == 5) __builtin_amdgcn_sched_barrier(0);          // (all SPT adds scheduled at once: their addresses and answers spill)
    }
    __syncthreads();
    DD_PHASE(3);          // entries read, groups counted
    if (wave == 0) {
        constexpr int PER = NB / 64;
        u32 cs[PER], sum = 0;
#pragma unroll
        for (int r = 0; r < PER; r++) { cs[r] = sm.g.bc[PER * lane + r]; sum += cs[r]; }
        const u32 inc = wave_incl_scan_u32(sum);
        u32 run = inc - sum;
#pragma unroll
        for (int r = 0; r < PER; r++) { sm.g.bbase[PER * lane + r] = run; run += cs[r]; }
        if (lane == 63) sm.g.bbase[NB] = inc;
    }
    __syncthreads();
    const u32 total = sm.g.bbase[NB];
    if (tid == 0) a.nwords[chunk] = total;
    if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = sm.g.bbase[16 * tid + 16] - sm.g.bbase[16 * tid];          // sixteen groups = one 6-bit start (tag_bits >= 14)
    u16* cw16 = reinterpret_cast<u16*>(sm.cnt);
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        if (ec[j]) {
            const u32 p = sm.g.bbase[(et[j] >> bsh) & (NB - 1)] + (ec[j] >> 16);
            sm.keys[p] = et[j];
            cw16[p] = (u16)ec[j];
        }
    }
    __syncthreads();
    DD_PHASE(4);          // grouped
    next_block(nchunk);          // the next block's first tile travels while this one is ranked and written (asked for any earlier, its
                                 // registers are held beside the entries above: 20 spilled)
    const u64 hi_part = (u64)chunk << a.tag_bits;          // the bits every key of the block has above its tag
    // three entries of a thread at a time: their LDS round trips (entry, group bounds, the group's first four tags) overlap -- one entry
    // after the other this loop was a chain of five round trips per entry, 11 K of a block's 52 K cycles
    constexpr int E = 3;
    for (u32 i0 = (u32)tid; i0 < total; i0 += E * BLOCK) {
        u32 mine[E], cc[E], g0[E], g1[E], rank[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            const u32 i = i0 + e * BLOCK;
            mine[e] = sm.keys[i < total ? i : 0u];
            cc[e] = cnt16[i < total ? i : 0u];
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const u32 b = (mine[e] >> bsh) & (NB - 1);
            g0[e] = sm.g.bbase[b]; g1[e] = sm.g.bbase[b + 1];
        }
        u32 o[E][4];
#pragma unroll
        for (int e = 0; e < E; e++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[e][r] = sm.keys[g0[e] + r < (u32)ALL ? g0[e] + r : (u32)ALL - 1];
#pragma unroll
        for (int e = 0; e < E; e++) {
            rank[e] = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) rank[e] += (g0[e] + r < g1[e] && o[e][r] < mine[e]) ? 1u : 0u;
            for (u32 q = g0[e] + 4; q < g1[e]; q += 4) {          // (a group of more than four: rare with 1024 groups of ~3)
                u32 p4[4];
#pragma unroll
                for (int r = 0; r < 4; r++) p4[r] = sm.keys[q + r < (u32)ALL ? q + r : (u32)ALL - 1];
#pragma unroll
                for (int r = 0; r < 4; r++) rank[e] += (q + r < g1[e] && p4[r] < mine[e]) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            const u32 i = i0 + e * BLOCK;
            if (i >= total) break;
            const u32 c = cc[e];
            const u64 k = hi_part | (u64)mine[e];
            if (c > maxc) {
                const u32 at = atomicAdd(a.n_big, 1u);
                if (at < a.big_cap) { a.big[2 * (u64)at] = k; a.big[2 * (u64)at + 1] = c; }
                atomicOr(a.flags, 2u);
            }
            a.out[lo + g0[e] + rank[e]] = (k << a.pack) | (u64)(c > maxc ? 0u : c);
        }
    }
    DD_PHASE(5);          // ranked and written
}

// Persistent: two workgroups per CU draw the blocks from a counter -- in order, not strided: the sizes go with the first bases, a
// stride of the grid would give one workgroup all the big ones.
template <int BLOCK, int BATCH, int ITEMS, bool TAGIN>
__global__ __launch_bounds__(BLOCK, 2 * BLOCK / 256) void dedupe2_kernel(DedupeArgs a) {
    using S = Dedupe2Smem<BLOCK, BATCH, ITEMS>;
    __shared__ S sm;
    static_assert(sizeof(S) <= 80 * 1024, "two workgroups per CU");
    Dedupe2Next<ITEMS> st;
    if (threadIdx.x == 0) sm.ticket = atomicAdd(a.counter, 1u);
    __syncthreads();
    st.chunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    st.lo = 0; st.len = 0;
    const u32 tmask = a.tag_bits >= 32 ? ~0u : (1u << a.tag_bits) - 1u;
    if (st.chunk < a.chunks) {
        const u64 nlo = a.cuts[st.chunk], nhi = a.cuts[st.chunk + 1];
        st.lo = nlo;
        st.len = nhi - nlo < (u64)a.limit ? (u32)(nhi - nlo) : ~0u;
    }
    if (st.len - 1u < ~0u - 1u) dedupe2_load<BLOCK, ITEMS, TAGIN>(a, st.lo, 0, st.len, tmask, st.tag);
    __syncthreads();          // the ticket word is free again
    u32 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 tlast = a.dbg ? (u32)__builtin_amdgcn_s_memtime() : 0u;
    (void)tlast;
    u32 nblk = 0;
    while (st.chunk < a.chunks) {
        dedupe2_block<BLOCK, BATCH, ITEMS, TAGIN>(a, sm, st, ph, tlast);          // leaves the next block in st
        __syncthreads();          // the table and the ticket word are free again
        DD_PHASE(6);
        nblk++;
    }
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 256) {
        for (int k = 0; k < 8; k++) a.dbg[(u64)blockIdx.x * 16 + k] = ph[k];
        a.dbg[(u64)blockIdx.x * 16 + 8] = nblk;
    }
}

// variant: bits 0-1 = compare-and-swaps in flight per thread (0: 4, 1: 1, 2: 2).  Tiles of 8 tags a thread; 16 (twice the bytes on
// their way from memory): 14.5 against 12.6 ms.  (1024-thread workgroups, two to a CU, have 64 registers a thread: the entries a
// thread holds while the block is sorted do not fit -- 25 to 40 spilled; not instantiated.)
int launch_dedupe2(zk_ctx* c, const DedupeArgs& a, bool tagin, int variant) {
    const u32 want = 2u * (u32)c->num_cus;
    const u32 grid = a.chunks < want ? a.chunks : want;
#define ZK_DD2(N, T) hipLaunchKernelGGL((dedupe2_kernel<512, N, 8, T>), dim3(grid), dim3(512), 0, c->stream, a)
#define ZK_DD2B(T) switch (variant & 3) { case 1: ZK_DD2(1, T); break; case 2: ZK_DD2(2, T); break; default: ZK_DD2(4, T); break; }
    if (tagin) ZK_DD2B(true) else ZK_DD2B(false)
#undef ZK_DD2B
#undef ZK_DD2
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
