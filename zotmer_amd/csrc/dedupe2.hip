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

namespace zk {

template <int BLOCK_>
struct Dedupe2Smem {
    static constexpr int BLOCK = BLOCK_, ITEMS = 8, TILE = BLOCK * ITEMS, NW = BLOCK / 64, ALL = 11264, SPT = ALL / BLOCK, NB = 256, SIDE = 128;
    static_assert(ALL % BLOCK == 0 && ALL % 8 == 0, "whole rounds, whole quads");
    u32 keys[ALL];            // tags (after the count: the entries again, grouped by their top byte)
    u32 cnt[ALL / 2];         // 16-bit counts: entry h in half (h & 1) of word h >> 1
    u32 side[NW][SIDE];
    u32 bc[NB];               // entries per top byte of the tag
    u32 bbase[NB + 1];        // ... before it
    u32 ticket;
};

// what a workgroup carries from one block to the next: the block it is about to count, with the first tile of its tags already
// asked for -- the ticket, the bounds and those tags travel while the previous block is being sorted and written
template <int ITEMS>
struct Dedupe2Next {
    u32 chunk;
    u64 lo, hi;
    u32 tag[ITEMS];
};

template <int BLOCK, bool TAGIN, bool RDFIRST>
__device__ __forceinline__ void dedupe2_block(const DedupeArgs& a, Dedupe2Smem<BLOCK>& sm, Dedupe2Next<8>& st, u32 (&ph)[8], u32& tlast) {
    using S = Dedupe2Smem<BLOCK>;
    constexpr int ITEMS = S::ITEMS, TILE = S::TILE, ALL = S::ALL, SPT = S::SPT, NB = S::NB;
    constexpr u32 EMPTY = ~0u;            // no entry; the all-ones tag has the last entry to itself (see `home`)
    constexpr u32 HS = ALL - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 chunk = st.chunk;
    const u64 lo = st.lo, hi = st.hi;
    const u32 maxc = (1u << a.pack) - 1u;
    const u32 tmask = a.tag_bits >= 32 ? ~0u : (1u << a.tag_bits) - 1u;
    // A whole tile: every thread takes ITEMS tags with 16-byte loads (TAGIN; a block starts wherever it starts: 4-byte aligned, no
    // more) or ITEMS keys a stride apart.  The cut last tile: one element a stride apart each, the ones beyond the block not loaded.
    // Which thread takes which key is the table's business alone.
    auto load = [&](u64 base, u64 end, u32 (&t)[ITEMS]) {
        if constexpr (TAGIN) {
            if (base + TILE <= end) {
                struct __attribute__((packed, aligned(4))) Tag4 { u32 a, b, c, d; };
                static_assert(ITEMS % 4 == 0, "whole quads");
#pragma unroll
                for (int i = 0; i < ITEMS / 4; i++) {
                    const Tag4 q = *reinterpret_cast<const Tag4*>(a.tin + base + (u64)i * (4 * BLOCK) + 4 * tid);
                    t[4 * i] = q.a; t[4 * i + 1] = q.b; t[4 * i + 2] = q.c; t[4 * i + 3] = q.d;
                }
            } else {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) {
                    const u64 g = base + (u64)i * BLOCK + tid;
                    t[i] = g < end ? a.tin[g] : 0u;
                }
            }
        } else if (base + TILE <= end) {
            const u64* p = a.kin + base + tid;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) t[i] = (u32)p[i * BLOCK] & tmask;
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u64 g = base + (u64)i * BLOCK + tid;
                t[i] = g < end ? (u32)a.kin[g] & tmask : 0u;
            }
        }
    };
    auto next_block = [&]() {          // call after a barrier that follows the ticket's store
        st.chunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
        st.lo = st.hi = 0;
        if (st.chunk < a.chunks) { st.lo = a.cuts[st.chunk]; st.hi = a.cuts[st.chunk + 1]; }
        if (st.hi > st.lo && st.hi - st.lo < (u64)a.limit) load(st.lo, st.hi, st.tag);
    };
    if (tid == 0) sm.ticket = atomicAdd(a.counter, 1u);          // the block after this one: read after the next barrier
    if (hi <= lo || hi - lo >= (u64)a.limit) {
        // nothing to count -- or more keys than a 16-bit count is safe for: dedupe_kernel's business
        if (tid == 0) {
            if (hi > lo) a.retry[atomicAdd(a.n_retry, 1u)] = chunk;
            a.nwords[chunk] = 0;
        }
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        __syncthreads();
        next_block();
        return;
    }
    for (int q = tid; q < ALL / 4; q += BLOCK) reinterpret_cast<uint4*>(sm.keys)[q] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int q = tid; q < ALL / 8; q += BLOCK) reinterpret_cast<uint4*>(sm.cnt)[q] = make_uint4(0, 0, 0, 0);
    if (tid < NB) sm.bc[tid] = 0;
    __syncthreads();
    DD_PHASE(0);          // table cleared
    const u32 nchunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    u32 bad = 0;
    u32 nside = 0;          // entries in this wave's side list (the same in every lane)
    auto home = [&](u32 e) -> u32 {
        // the all-ones tag (= the empty marker) has the last entry to itself: there the swap of "empty" for "empty" succeeds and
        // leaves the word as it is; no other key is ever sent there
        const u32 x = e * 0x9E3779B1u;
        return e == EMPTY ? HS : (u32)(((u64)x * HS) >> 32);
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
    // The common case has no loop and no branch: one compare-and-swap at the tag's home entry (RDFIRST: a plain read, the swap only
    // for the lanes that see "empty"), the count added as 1 or 0 (adding 0 to another key's entry harms nobody); a key that finds
    // another key at home goes to the wave's side list (its place from a ballot, no atomic).
    auto insert = [&](u32 e, bool valid) {
        const u32 h = home(e);
        u32 old;
        if constexpr (RDFIRST) {
            old = sm.keys[h];
            if (old == EMPTY && valid) old = atomicCAS(&sm.keys[h], EMPTY, e);
            if (!valid) old = e;
        } else old = valid ? atomicCAS(&sm.keys[h], EMPTY, e) : e;
        const bool ok = old == EMPTY || old == e;
        count(h, (ok && valid) ? 1u : 0u);
        const u64 m = __ballot(!ok);
        if (m) {
            if (!ok) sm.side[wave][nside + popc_below(m)] = e;
            nside += (u32)__popcll(m);
            if (nside >= 64) { nside -= 64; drain(nside, 64); }
        }
    };
    u32 tag[ITEMS], nt[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) tag[i] = st.tag[i];          // the first tile was asked for during the previous block
    for (u64 base = lo; base < hi; base += TILE) {
        if (base + TILE < hi) load(base + TILE, hi, nt);
        if (base + TILE <= hi) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) insert(tag[i], true);
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) insert(tag[i], base + (u64)i * BLOCK + tid < hi);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) tag[i] = nt[i];
    }
    drain(0, nside);
    DD_PHASE(1);          // keys inserted
    st.chunk = nchunk; st.lo = st.hi = 0;
    if (nchunk < a.chunks) { st.lo = a.cuts[nchunk]; st.hi = a.cuts[nchunk + 1]; }
    if (st.hi > st.lo && st.hi - st.lo < (u64)a.limit) load(st.lo, st.hi, st.tag);          // the next block's first tile travels while this one is sorted and written
    const int any_bad = __syncthreads_or((int)bad);
    DD_PHASE(2);          // ... every wave done
    if (any_bad) {
        // the table filled up (a block with more distinct keys than it holds): dedupe_kernel's larger table gets a try
        if (tid == 0) {
            a.retry[atomicAdd(a.n_retry, 1u)] = chunk;
            a.nwords[chunk] = 0;
        }
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        return;
    }
    // ---- the block's entries, sorted: a counting sort on the tag's top byte, then ranks inside each byte's group ---------
    // thread t takes the entries t, t + BLOCK, ... into registers with their places in their byte groups (a returning add); the
    // table's memory then takes them back grouped
    u32 et[SPT], ec[SPT];          // ec: count | place in the group << 16
    const int bsh = a.tag_bits > 8 ? a.tag_bits - 8 : 0;
    const u16* cnt16 = reinterpret_cast<const u16*>(sm.cnt);
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        et[j] = sm.keys[tid + j * BLOCK];
        ec[j] = cnt16[tid + j * BLOCK];
    }
#pragma unroll
    for (int j = 0; j < SPT; j++)
        if (ec[j]) ec[j] |= atomicAdd(&sm.bc[(et[j] >> bsh) & (NB - 1)], 1u) << 16;
    __syncthreads();
    DD_PHASE(3);          // entries read, byte groups counted
    if (wave == 0) {
        u32 c4[4], sum = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) { c4[r] = sm.bc[4 * lane + r]; sum += c4[r]; }
        const u32 inc = wave_incl_scan_u32(sum);
        u32 run = inc - sum;
#pragma unroll
        for (int r = 0; r < 4; r++) { sm.bbase[4 * lane + r] = run; run += c4[r]; }
        if (lane == 63) sm.bbase[NB] = inc;
    }
    __syncthreads();
    const u32 total = sm.bbase[NB];
    if (tid == 0) a.nwords[chunk] = total;
    if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = sm.bbase[4 * tid + 4] - sm.bbase[4 * tid];          // four top bytes = one 6-bit start
    u16* cw16 = reinterpret_cast<u16*>(sm.cnt);
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        if (ec[j]) {
            const u32 p = sm.bbase[(et[j] >> bsh) & (NB - 1)] + (ec[j] >> 16);
            sm.keys[p] = et[j];
            cw16[p] = (u16)ec[j];
        }
    }
    __syncthreads();
    DD_PHASE(4);          // grouped by top byte
    const u64 hi_part = (u64)chunk << a.tag_bits;          // the bits every key of the block has above its tag
    for (u32 i = (u32)tid; i < total; i += BLOCK) {
        const u32 mine = sm.keys[i];
        const u32 b = (mine >> bsh) & (NB - 1);
        const u32 g0 = sm.bbase[b], g1 = sm.bbase[b + 1];
        u32 rank = 0;
        for (u32 q = g0; q < g1; q++) rank += sm.keys[q] < mine ? 1u : 0u;
        const u32 c = cnt16[i];
        const u64 k = hi_part | (u64)mine;
        if (c > maxc) {
            const u32 at = atomicAdd(a.n_big, 1u);
            if (at < a.big_cap) { a.big[2 * (u64)at] = k; a.big[2 * (u64)at + 1] = c; }
            atomicOr(a.flags, 2u);
        }
        a.out[lo + g0 + rank] = (k << a.pack) | (u64)(c > maxc ? 0u : c);
    }
    DD_PHASE(5);          // ranked and written
}

// Persistent: two workgroups per CU draw the blocks from a counter -- in order, not strided: the sizes go with the first bases, a
// stride of the grid would give one workgroup all the big ones.
template <int BLOCK, bool TAGIN, bool RDFIRST>
__global__ __launch_bounds__(BLOCK, 2 * BLOCK / 256) void dedupe2_kernel(DedupeArgs a) {
    using S = Dedupe2Smem<BLOCK>;
    __shared__ S sm;
    static_assert(sizeof(S) <= 80 * 1024, "two workgroups per CU");
    Dedupe2Next<S::ITEMS> st;
    if (threadIdx.x == 0) sm.ticket = atomicAdd(a.counter, 1u);
    __syncthreads();
    st.chunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    st.lo = st.hi = 0;
    if (st.chunk < a.chunks) { st.lo = a.cuts[st.chunk]; st.hi = a.cuts[st.chunk + 1]; }
    const u32 tmask = a.tag_bits >= 32 ? ~0u : (1u << a.tag_bits) - 1u;
#pragma unroll
    for (int i = 0; i < S::ITEMS; i++) {
        const u64 g = st.lo + (u64)i * BLOCK + threadIdx.x;
        if constexpr (TAGIN) st.tag[i] = g < st.hi ? a.tin[g] : 0u;
        else st.tag[i] = g < st.hi ? (u32)a.kin[g] & tmask : 0u;
    }
    __syncthreads();          // the ticket word is free again
    u32 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 tlast = a.dbg ? (u32)__builtin_amdgcn_s_memtime() : 0u;
    (void)tlast;
    u32 nblk = 0;
    while (st.chunk < a.chunks) {
        dedupe2_block<BLOCK, TAGIN, RDFIRST>(a, sm, st, ph, tlast);          // leaves the next block in st
        __syncthreads();          // the table and the ticket word are free again
        DD_PHASE(6);
        nblk++;
    }
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 256) {
        for (int k = 0; k < 8; k++) a.dbg[(u64)blockIdx.x * 16 + k] = ph[k];
        a.dbg[(u64)blockIdx.x * 16 + 8] = nblk;
    }
}

// variant: bit 1 = a plain read before the compare-and-swap.  (1024-thread workgroups, two to a CU, have 64 registers a thread: the
// entries a thread holds while the block is sorted do not fit -- 25 to 40 spilled; not instantiated.)
int launch_dedupe2(zk_ctx* c, const DedupeArgs& a, bool tagin, int variant) {
    const u32 want = 2u * (u32)c->num_cus;
    const u32 grid = a.chunks < want ? a.chunks : want;
#define ZK_DD2(B, T, R) hipLaunchKernelGGL((dedupe2_kernel<B, T, R>), dim3(grid), dim3(B), 0, c->stream, a)
    switch ((variant & 2) | (tagin ? 4 : 0)) {
        case 0: ZK_DD2(512, false, false); break;
        case 2: ZK_DD2(512, false, true); break;
        case 4: ZK_DD2(512, true, false); break;
        default: ZK_DD2(512, true, true); break;
    }
#undef ZK_DD2
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
