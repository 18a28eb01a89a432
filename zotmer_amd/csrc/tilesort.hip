// tilesort.hip -- K2b, finishing a radix sort in LDS.
//
// Replaces the lower passes of numpy's sort in kmerize (zotmer/commands/kmerize.py:41-132, `xs.sort()`; library/misc.py:400-424 for the
// pair lists) exactly as the LSD passes of radix_sort.hip do: same keys in, same sorted keys out.
//
// An LSD sort moves every key once per digit: 6 passes of 16 bytes a key at K = 25, 7 at K = 31.  But once the TOP bits are sorted -- by
// the same stable passes, taken over the bits [key_bits - t, key_bits) only -- the array is a row of blocks of equal top bits in key
// order, and as soon as a block is a few dozen keys the rest of the sort is local: cut the array into tiles of about T keys AT BLOCK
// BORDERS (tile_bounds_kernel: a binary search over the S keys before every multiple of T), and a workgroup sorts a tile in LDS all the
// way, whatever number of bits is left, at 16 bytes a key ONCE.
//
// In LDS (tile_sort_kernel) a tile is sorted the way dedupe2_kernel sorts a block's entries: a counting sort into G = 4096 (2048) groups by
// the key's place in the tile's value range (the tile spans whole blocks, so the range is known from its first and last key's top bits;
// the keys of a tile are spread evenly enough over it for groups of ~1.5), then every key's rank inside its group by comparing -- the
// key's own LDS index breaks ties, so equal keys get different places -- and the store goes straight to its final place: the 64 lanes
// of a wave hold neighbours of the grouped order, so what they write is a permutation of one contiguous 512 bytes.
//
// Exact for any input: the result never depends on how the keys are spread (a crowded group only costs compares), and an input with
// a block of more than S keys -- more than S copies of one top-bit pattern: low-complexity sequence in reads that otherwise do not
// repeat -- raises a flag instead of being cut wrongly; the caller then sorts the remaining bits the long way (the array is untouched
// where a tile was too large, a permutation of the input everywhere).
#include "internal.hpp"

// in-kernel s_memtime per phase of a tile (diagnostic build, tools/build_phases.sh; tools/ts_phases.py reads the sums)
#ifdef ZK_PHASES
#define TS_PHASE(k) do { if (a.dbg) { const u32 now__ = (u32)__builtin_amdgcn_s_memtime(); ph[k] += now__ - tlast; tlast = now__; } } while (0)
#else
#define TS_PHASE(k) do { } while (0)
#endif

namespace zk {

constexpr int TS_BLOCK = 512;
constexpr u32 TS_SLACK = 1024;          // S: no block of equal top bits may be longer

// PAIRS: a 32-bit payload travels with every key, and equal keys keep the order they came in (zk_sort_pairs is stable): the
// place a pair had in the tile is kept beside it and breaks the ties (keys alone need no such thing: equal keys are the same key)
template <int ITEMS, int GROUPS, bool PAIRS>
struct TileSortSmem {
    static constexpr int CAP = TS_BLOCK * ITEMS;
    alignas(16) u64 keys[CAP];
    u32 vals[PAIRS ? CAP : 1];
    u16 idx[PAIRS ? CAP : 1];
    alignas(16) u32 start[GROUPS + 8];          // the groups' counts, then where they start (start[G] = the tile's keys)
    u32 wsum[TS_BLOCK / 64];
};
typedef TileSortSmem<14, 4096, false> TileSortKeys;
constexpr int TS_COUNT_ITEMS = 12;          // the counting form holds a tile's entries in registers over the next tile's load: 12 a thread fit
typedef TileSortSmem<10, 2048, true> TileSortPairs;

struct TileSortArgs {
    const u64* kin; u64* kout;
    const u32* vin; u32* vout;
    const u64* bounds;          // [tiles + 1]
    u32 tiles;
    int pshift;                 // the bits from here up are sorted already
};

// bounds[t] = the first key of the block that the key at t * T belongs to (t = 1 .. tiles - 1); bounds[0] = 0, bounds[tiles] = n
__global__ void tile_bounds_kernel(const u64* __restrict__ a, u64 n, int pshift, u32 T, u32 S, u32 tiles, u64* __restrict__ bounds,
                                   u32* __restrict__ flag) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles) return;
    if (t == 0) { bounds[0] = 0; return; }
    if (t == tiles) { bounds[tiles] = n; return; }
    const u64 p = (u64)t * T;          // < n, and >= T > S
    const u64 pre = a[p] >> pshift;
    u64 lo = p - S, hi = p;            // a[hi] has the prefix; is a[lo]'s smaller?
    if ((a[lo] >> pshift) == pre) {
        atomicOr(flag, 1u);            // a block of more than S keys: not cut here (the tile is then too large and skipped)
        bounds[t] = lo;
        return;
    }
    while (hi - lo > 1) {              // invariant: prefix(a[lo]) < pre == prefix(a[hi])
        const u64 mid = lo + (hi - lo) / 2;
        if ((a[mid] >> pshift) == pre) hi = mid; else lo = mid;
    }
    bounds[t] = hi;
}

// the map key -> group of a tile: monotone, g = floor(d * G / ((range >> sh) + 1)) for d = (key - first) >> sh, in 32 bits
struct TileMap {
    u64 kmin; int sh; u32 scale;
    __device__ __forceinline__ u32 group(u64 key) const {
        const u32 d = (u32)((key - kmin) >> sh);
        return scale ? __umulhi(d, scale) : d;
    }
};

// The tile [lo, lo + m) of kin (and vin) into LDS, grouped: sm.keys (vals, idx) hold the tile's entries group by group, sm.start where
// every group starts (start[G] = m).  Ends with a barrier.
template <int ITEMS, int G, bool PAIRS, class S>
__device__ __forceinline__ TileMap tile_group(S& sm, const u64* kin, const u32* vin, u64 lo, u32 m, int pshift, int tid) {
    constexpr int NW = TS_BLOCK / 64, QPT = G / TS_BLOCK / 4;          // QPT: quads of groups a thread scans
    static_assert(QPT == 1 || QPT == 2, "four or eight groups a thread in the scan");
    static_assert(S::CAP <= 8192 && G <= 4096, "group | place << 12 in a word");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64* kp = kin + lo;
    u64 k[ITEMS];
    u32 v[PAIRS ? ITEMS : 1];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const u32 i = (u32)j * TS_BLOCK + tid;
        k[j] = i < m ? kp[i] : ~0ull;
        if constexpr (PAIRS) v[j] = i < m ? vin[lo + i] : 0u;
    }
    // the tile's value range, from the top bits of its first and last key (whole blocks)
    const u64 first = kp[0], last = kp[m - 1];
    TileMap tm;
    tm.kmin = (first >> pshift) << pshift;
    const u64 rm1 = (((last >> pshift) - (first >> pshift)) << pshift) | ((1ull << pshift) - 1ull);
    tm.sh = rm1 >> 32 ? 32 - __builtin_clzll(rm1) : 0;
    const u32 rs = (u32)(rm1 >> tm.sh);
    tm.scale = rs < (u32)G ? 0u : (u32)(((u64)G << 32) / ((u64)rs + 1ull));
    {
        uint4* z = reinterpret_cast<uint4*>(sm.start);
#pragma unroll
        for (int q = 0; q < QPT; q++) z[QPT * tid + q] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    u32 gp[ITEMS];          // group | place in the group << 12
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const u32 i = (u32)j * TS_BLOCK + tid;
        const u32 g = i < m ? tm.group(k[j]) : 0u;
        gp[j] = g;
        if (i < m) gp[j] |= atomicAdd(&sm.start[g], 1u) << 12;
    }
    __syncthreads();
    {
        // counts -> starts: four or eight groups a thread, the waves' sums through LDS
        uint4* z = reinterpret_cast<uint4*>(sm.start);
        uint4 cq[QPT];
        u32 sum = 0;
#pragma unroll
        for (int q = 0; q < QPT; q++) { cq[q] = z[QPT * tid + q]; sum += cq[q].x + cq[q].y + cq[q].z + cq[q].w; }
        const u32 inc = wave_incl_scan_u32(sum);
        if (lane == 63) sm.wsum[wave] = inc;
        __syncthreads();
        u32 run = inc - sum;
#pragma unroll
        for (int w = 0; w < NW; w++) run += w < wave ? sm.wsum[w] : 0u;
#pragma unroll
        for (int q = 0; q < QPT; q++) {
            uint4 sq;
            sq.x = run; run += cq[q].x; sq.y = run; run += cq[q].y; sq.z = run; run += cq[q].z; sq.w = run; run += cq[q].w;
            z[QPT * tid + q] = sq;
        }
        if (tid == 0) sm.start[G] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const u32 i = (u32)j * TS_BLOCK + tid;
        if (i < m) {
            const u32 p = sm.start[gp[j] & (G - 1)] + (gp[j] >> 12);
            sm.keys[p] = k[j];
            if constexpr (PAIRS) { sm.vals[p] = v[j]; sm.idx[p] = (u16)i; }
        }
    }
    __syncthreads();
    return tm;
}

// The final places of E grouped entries (i[e] < m or not: entries beyond the tile get a place nobody uses): their group's start plus
// the entries of the group that go before them.  The E entries' LDS round trips (entry, group bounds, the group's first four keys)
// overlap.
template <int E, bool PAIRS, class S>
__device__ __forceinline__ void tile_rank(const S& sm, const TileMap& tm, const u32 (&i)[E], u32 m, u64 (&mine)[E], u32 (&place)[E]) {
    constexpr u32 CAP = S::CAP;
    // does the entry at q (key ko) go before the one at at (key km)?  Equal keys: the one that came first (pairs), any fixed order (keys)
    auto before = [&](u64 ko, u32 q, u64 km, u32 at) -> bool {
        if (ko != km) return ko < km;
        if constexpr (PAIRS) return q != at && sm.idx[q] < sm.idx[at];          // (rare: the two extra reads are taken by the lanes that need them)
        else return q < at;
    };
    u32 g0[E], g1[E];
#pragma unroll
    for (int e = 0; e < E; e++) mine[e] = sm.keys[i[e] < m ? i[e] : 0u];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const u32 g = tm.group(mine[e]);
        g0[e] = sm.start[g]; g1[e] = sm.start[g + 1];
    }
    u64 o[E][4];
#pragma unroll
    for (int e = 0; e < E; e++)
#pragma unroll
        for (int r = 0; r < 4; r++) o[e][r] = sm.keys[g0[e] + r < CAP ? g0[e] + r : CAP - 1];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const u32 at = i[e] < m ? i[e] : 0u;
        u32 rank = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u32 q = g0[e] + r;
            rank += (q < g1[e] && before(o[e][r], q, mine[e], at)) ? 1u : 0u;
        }
        for (u32 q0 = g0[e] + 4; q0 < g1[e]; q0 += 4) {          // (a group of more than four)
            u64 p4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) p4[r] = sm.keys[q0 + r < CAP ? q0 + r : CAP - 1];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const u32 q = q0 + r;
                rank += (q < g1[e] && before(p4[r], q, mine[e], at)) ? 1u : 0u;
            }
        }
        place[e] = g0[e] + rank;
    }
}

template <int ITEMS, int G, bool PAIRS>
__global__ __launch_bounds__(TS_BLOCK, 2 * TS_BLOCK / 256) void tile_sort_kernel(TileSortArgs a) {
    using S = TileSortSmem<ITEMS, G, PAIRS>;
    constexpr int CAP = S::CAP;
    static_assert(sizeof(S) <= 80 * 1024, "two workgroups per CU");
    __shared__ S sm;
    for (u32 t = blockIdx.x; t < a.tiles; t += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));          // (opaque per tile: nothing derived from it is kept, as a 64-bit pair, across tiles)
        const u64 lo = a.bounds[t], hi = a.bounds[t + 1];
        if (hi - lo > (u64)CAP || hi == lo) continue;          // too large: flagged by tile_bounds_kernel, the caller sorts the long way
        const u32 m = (u32)(hi - lo);
        const TileMap tm = tile_group<ITEMS, G, PAIRS>(sm, a.kin, a.vin, lo, m, a.pshift, tid);
        // every entry straight to its final place: the 64 lanes of a wave hold neighbours of the grouped order
        constexpr int E = 2;
        u64* op = a.kout + lo;
        for (u32 i0 = (u32)tid; i0 < m; i0 += E * TS_BLOCK) {
            u32 i[E], place[E];
            u64 mine[E];
#pragma unroll
            for (int e = 0; e < E; e++) i[e] = i0 + e * TS_BLOCK;
            tile_rank<E, PAIRS>(sm, tm, i, m, mine, place);
#pragma unroll
            for (int e = 0; e < E; e++) {
                if (i[e] >= m) break;
                op[place[e]] = mine[e];
                if constexpr (PAIRS) a.vout[lo + place[e]] = sm.vals[i[e]];
            }
        }
        __syncthreads();          // the tile's LDS is free again
    }
}

// ---- ... and counted: the distinct keys of the sorted array with their numbers of copies, without the sorted array ever being written ----
//
// Equal keys share all their bits, so they sit in one tile, and a tile sorted in LDS can be run-length counted there: heads flagged,
// their places compacted (into the memory of the group table, which is done with), a decoupled look-back over the tiles -- numbered
// in the order the workgroups draw them -- says where the tile's distinct keys go, and they leave with their counts, 12 bytes per
// DISTINCT key.  uniq may be the input array itself (a tile learns its place only after every earlier tile has loaded its keys, and
// no tile writes beyond its own input).
struct TileCountArgs {
    const u64* kin;
    const u64* bounds;
    u32 tiles;
    int pshift;
    u64* uniq; u32* counts; u64 cap;
    const u32* flag;          // set by tile_bounds_kernel: a block too long -- nothing is written at all
    u64* status; u32* ticket; u32 ticket_base, epoch; u32* err; u64* d_total;
    u64* dbg;                 // diagnostic build: where the phase sums go (else null)
};

template <int ITEMS, int G>
__global__ __launch_bounds__(TS_BLOCK, 2 * TS_BLOCK / 256) void tile_sort_count_kernel(TileCountArgs a) {
    using S = TileSortSmem<ITEMS, G, false>;
    constexpr int CAP = S::CAP, NW = TS_BLOCK / 64;
    static_assert(sizeof(S) <= 80 * 1024, "two workgroups per CU");
    static_assert(2 * (G + 8) >= CAP + 1, "the heads' places fit the group table's memory, 16 bits each");
    static_assert(ITEMS % 2 == 0, "two entries at a time");
    __shared__ S sm;
    __shared__ u32 s_ticket;
    __shared__ u64 s_base;
    if (*a.flag) {          // (the tickets this launch was given are drawn all the same: the next launch counts from there)
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.ticket, a.tiles + gridDim.x);
        return;
    }
    u32 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ntile = 0;
    u32 tlast = a.dbg ? (u32)__builtin_amdgcn_s_memtime() : 0u;
    (void)tlast; (void)ph;
    // A tile does not wait for its place among the distinct keys: it publishes its count, takes its (k-mer, count) entries into
    // registers and goes on to load and group the next tile; only then does it look back -- by then the tiles before it, which run
    // in step with it, have published theirs (waiting at once cost 29 % of a tile: every tile waited for the slowest of ~500).
    u64 hk[ITEMS];          // the held tile's entries tid, tid + BLOCK, ...
    u32 hc[ITEMS / 2];      // ... and their counts, two to a word
    bool have = false;
    u32 pt = 0, ptotal = 0;
    auto resolve_and_write = [&](int tid) {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (wave == 0) {
            const u64 ex = lookback_resolve(a.status, pt, ptotal, a.epoch, a.err);
            if (lane == 0) {
                s_base = ex;
                if (pt == a.tiles - 1) *a.d_total = ex + ptotal;
                if (ex + ptotal > a.cap) atomicOr(a.err, ZK_DERR_CAPACITY);
            }
        }
        __syncthreads();
        const u64 base = s_base;
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const u32 u = (u32)j * TS_BLOCK + tid;
            if (u < ptotal && base + u < a.cap) {
                a.uniq[base + u] = hk[j];
                a.counts[base + u] = (hc[j / 2] >> (16 * (j & 1))) & 0xffffu;
            }
        }
    };
    while (true) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const u32 t = take_ticket(a.ticket, &s_ticket) - a.ticket_base;          // (barrier inside)
        if (t >= a.tiles) break;
        const u64 lo = a.bounds[t], hi = a.bounds[t + 1];
        const u32 m = (u32)(hi - lo);          // 1 .. CAP: no block was too long
        TS_PHASE(0);          // ticket
        const TileMap tm = tile_group<ITEMS, G, false>(sm, a.kin, nullptr, lo, m, a.pshift, tid);
        TS_PHASE(1);          // loaded and grouped
        if (have) resolve_and_write(tid);          // the tile before: its place, its entries out of the registers
        TS_PHASE(5);          // look-back of the tile before (wave 0), barrier, its stores issued
        u64 mine[ITEMS];
        u32 place[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; j += 2) {
            u32 i[2] = {(u32)j * TS_BLOCK + tid, (u32)(j + 1) * TS_BLOCK + tid};
            u64 mm[2];
            u32 pp[2];
            tile_rank<2, false>(sm, tm, i, m, mm, pp);
            mine[j] = mm[0]; mine[j + 1] = mm[1];
            place[j] = pp[0]; place[j + 1] = pp[1];
        }
        TS_PHASE(2);          // ranked
        __syncthreads();          // every entry read: the tile goes back sorted
#pragma unroll
        for (int j = 0; j < ITEMS; j++)
            if ((u32)j * TS_BLOCK + tid < m) sm.keys[place[j]] = mine[j];
        __syncthreads();
        TS_PHASE(3);          // sorted in LDS
        // heads: thread t looks at the ITEMS consecutive entries from t * ITEMS
        u32 total = 0;
        u16* hp = reinterpret_cast<u16*>(sm.start);
        {
            const u32 p0 = (u32)tid * ITEMS;
            u64 prev = p0 > 0 && p0 <= m ? sm.keys[p0 - 1] : 0ull;
            u32 hm = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const u32 p = p0 + j;
                const u64 x = sm.keys[p < m ? p : 0u];
                if (p < m && (p == 0 || x != prev)) hm |= 1u << j;
                prev = x;
            }
            const u32 nh = (u32)__popc(hm);
            const u32 inc = wave_incl_scan_u32(nh);
            if (lane == 63) sm.wsum[wave] = inc;
            __syncthreads();          // (also: nobody reads the group table any more)
            u32 u = inc - nh;
#pragma unroll
            for (int w = 0; w < NW; w++) { u += w < wave ? sm.wsum[w] : 0u; total += sm.wsum[w]; }
#pragma unroll
            for (int j = 0; j < ITEMS; j++)
                if ((hm >> j) & 1u) hp[u++] = (u16)(p0 + j);
            if (tid == 0) hp[total] = (u16)m;
        }
        if (tid == 0) lookback_publish(a.status, t, total, a.epoch);
        __syncthreads();          // the heads' places are there
        TS_PHASE(4);          // heads
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const u32 u = (u32)j * TS_BLOCK + tid;
            const u32 p = hp[u < total ? u : 0u], q = hp[u < total ? u + 1 : 0u];
            hk[j] = sm.keys[p];
            const u32 len = (q - p) & 0xffffu;
            if (j & 1) hc[j / 2] |= len << 16; else hc[j / 2] = len;
        }
        have = true; pt = t; ptotal = total;
        __syncthreads();          // the tile's LDS (and the ticket word) are free again
        TS_PHASE(6);          // entries taken
        ntile++;
    }
    if (have) resolve_and_write(threadIdx.x);
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 256) {
        for (int k = 0; k < 8; k++) a.dbg[(u64)blockIdx.x * 16 + k] = ph[k];
        a.dbg[(u64)blockIdx.x * 16 + 8] = ntile;
    }
}

// How many top bits the LSD passes must have sorted before tiles can be cut: whole passes of rbits, blocks of about 64 keys or fewer
// on average (the canonical k-mers' blocks that start with A are twice the mean; S = 1024 leaves room for the rest).  0 = not worth
// it (the passes left over would be fewer than two).
int tile_sort_top_bits(uint64_t n, int key_bits, int rbits) {
    if (n < (1ull << 16)) return 0;
    int passes = 1;
    while ((n >> (rbits * passes)) > 64 && rbits * passes < key_bits) passes++;
    const int top = rbits * passes;
    const int all = (key_bits + rbits - 1) / rbits;
    if (passes + 2 > all || top >= key_bits) return 0;
    return top;
}

// keys (and vals, or null) hold n keys ordered by the bits [key_bits - top, key_bits): sorts them by all their bits, in place.
// *declined = true: a block of equal top bits was too long, the array is a permutation of what it was (still ordered by the top bits
// where tiles were sorted, untouched elsewhere) and the caller must sort it another way.
int tile_sort(zk_ctx* c, u64* keys, u32* vals, uint64_t n, int key_bits, int top, bool* declined) {
    *declined = false;
    if (n == 0) return ZK_OK;
    const bool pairs = vals != nullptr;
    const u32 cap = pairs ? (u32)TileSortPairs::CAP : (u32)TileSortKeys::CAP;
    const u32 T = cap - TS_SLACK;
    const uint64_t tiles64 = div_up(n, T);
    if (tiles64 >= (1ull << 31)) return fail(c, ZK_EINVAL, "tile sort: %llu keys", (unsigned long long)n);
    const u32 tiles = (u32)tiles64;
    u64* bounds;
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)tiles + 1), (void**)&bounds));
    u32* flag = (u32*)(c->d_scalars + 40);
    ZK_HIP(c, hipMemsetAsync(flag, 0, sizeof(u64), c->stream));
    const int pshift = key_bits - top;
    hipLaunchKernelGGL(tile_bounds_kernel, dim3(tiles / 256 + 1), dim3(256), 0, c->stream, (const u64*)keys, (u64)n, pshift, T, TS_SLACK, tiles, bounds, flag);
    ZK_HIP(c, hipGetLastError());
    TileSortArgs a{keys, keys, vals, vals, bounds, tiles, pshift};
    const u32 want = 2u * (u32)c->num_cus;
    const u32 grid = tiles < want ? tiles : want;
    prof_begin(c, ZK_PROF_TILE_SORT, (pairs ? 24 : 16) * n);
    if (pairs) hipLaunchKernelGGL((tile_sort_kernel<10, 2048, true>), dim3(grid), dim3(TS_BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL((tile_sort_kernel<14, 4096, false>), dim3(grid), dim3(TS_BLOCK), 0, c->stream, a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 40, flag, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *declined = (c->h_scalars[40] & 1ull) != 0;
    return ZK_OK;
}

// ... and counted: keys hold n keys ordered by their top bits; uniq / counts (cap entries; uniq may be keys) get the distinct keys,
// ascending, and how often each occurs; *n_unique their number.  *declined: nothing was written, the caller must sort and count
// another way.
int tile_sort_count(zk_ctx* c, const u64* keys, uint64_t n, int key_bits, int top, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_unique,
                    bool* declined) {
    *declined = false;
    *n_unique = 0;
    if (n == 0) return ZK_OK;
    const u32 T = (u32)TS_BLOCK * TS_COUNT_ITEMS - TS_SLACK;
    const uint64_t tiles64 = div_up(n, T);
    if (tiles64 >= (1ull << 31)) return fail(c, ZK_EINVAL, "tile sort: %llu keys", (unsigned long long)n);
    const u32 tiles = (u32)tiles64;
    u64* bounds;
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)tiles + 1), (void**)&bounds));
    u32* flag = (u32*)(c->d_scalars + 40);
    ZK_HIP(c, hipMemsetAsync(flag, 0, sizeof(u64), c->stream));
    const int pshift = key_bits - top;
    hipLaunchKernelGGL(tile_bounds_kernel, dim3(tiles / 256 + 1), dim3(256), 0, c->stream, keys, (u64)n, pshift, T, TS_SLACK, tiles, bounds, flag);
    ZK_HIP(c, hipGetLastError());
    const u32 want = 2u * (u32)c->num_cus;
    const u32 grid = tiles < want ? tiles : want;
    TileCountArgs a{};
    a.kin = keys; a.bounds = bounds; a.tiles = tiles; a.pshift = pshift; a.uniq = uniq; a.counts = counts; a.cap = cap; a.flag = flag;
    // every workgroup draws tiles until the ticket says there are none left: tiles + grid tickets in all
    ZK_TRY(lookback_begin(c, tiles, tiles + grid, &a.epoch, &a.ticket_base));
    a.status = c->status; a.ticket = c->d_ticket; a.err = c->d_err; a.d_total = c->d_scalars + 9;
    a.dbg = c->dbg;
    prof_begin(c, ZK_PROF_TILE_SORT, 8 * n);
    hipLaunchKernelGGL((tile_sort_count_kernel<TS_COUNT_ITEMS, 4096>), dim3(grid), dim3(TS_BLOCK), 0, c->stream, a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 40, flag, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *declined = (c->h_scalars[40] & 1ull) != 0;
    if (*declined) return ZK_OK;
    *n_unique = c->h_scalars[9];
    prof_add_bytes(c, ZK_PROF_TILE_SORT, 12 * *n_unique);
    ZK_TRY(check_device_error(c));
    return ZK_OK;
}

}  // namespace zk
