// tag_pass.hip -- the SECOND pass of the two-pass plan of zk_kmerize, as a count / scan / scatter pass over static segments.
//
// Replaces (with stream_pass.hip's first pass) misc.radix_sort (zotmer/library/misc.py:400-424) over the k-mers of the reads
// (zotmer/commands/kmerize.py:412-417) as far as the block dedupe needs them ordered: by their top two digits.
//
// Pass 0 (stream_pass.hip, PLANES) leaves the keys ordered by digit d0 (bits 32-40 at K = 25) as two arrays: tag[i] = the key's low
// 32 bits, dig[i] = its next digit d1 (bits 41-49) -- 6 bytes a key; d0 is said by the key's place.  This pass orders them by d1,
// keeping the d0 order inside: block (d1, d0) of the block dedupe ends up contiguous, blocks in key order, and only the TAGS are
// written (the dedupe's table holds nothing else; the block's number is the place).  The look-back pipeline of radix_sort.hip did
// this in 25 ms: 8-byte keys read, 64-byte runs of tags written wherever the tiles' counts put them (WRITE_SIZE 1.56 x the tags).
// Here, as in pass 0:
//   * the input is cut into SEGMENTS, contiguous and inside ONE d0 bucket each (so a tile never mixes two d0 values and the order
//     inside a tile is free);
//   * a count kernel reads the digit array alone (2 bytes a key) for every segment's digit counts; a scan gives every
//     (segment, digit) its own contiguous piece of the output -- and the blocks' starts fall out of the same table;
//   * the scatter kernel walks its segment tile by tile, parks the tags grouped by digit in LDS and writes only WHOLE 64-byte units
//     (16 tags, aligned in the output); what is left of a digit (fewer than 16) waits in LDS for the next tile.
// No look-back, no scanner workgroups, no status words, no atomics outside LDS.
#include "internal.hpp"

#include <type_traits>

namespace zk {

constexpr int T1_BLOCK = 512, T1_RBITS = 9, T1_RADIX = 1 << T1_RBITS, T1_G = 16;

struct TagSegs {
    u64* start;        // [max_segs + 1] where segment s begins in the input (start[nseg] = n)
    u32* nseg;         // how many there are
    u32* first;        // [radix + 1] first[b] = segments in the buckets below b (= the first segment of bucket b, if it has any)
    u32 max_segs;
    u64 seg_len;
};

// One workgroup: bucket b = [ghist0[b], ghist0[b + 1]) is cut into segments of seg_len keys (the last one shorter).
__global__ __launch_bounds__(T1_RADIX) void tagpass_plan_kernel(const u64* __restrict__ ghist0, u64 n, u32 radix, TagSegs sg) {
    __shared__ u32 scan[T1_RADIX];
    const u32 b = threadIdx.x;
    const u64 lo = b < radix ? ghist0[b] : n, hi = b + 1 < radix ? ghist0[b + 1] : n;
    const u32 cnt = b < radix ? (u32)((hi - lo + sg.seg_len - 1) / sg.seg_len) : 0u;
    scan[b] = cnt;
    __syncthreads();
    for (u32 o = 1; o < (u32)T1_RADIX; o <<= 1) {
        const u32 v = b >= o ? scan[b - o] : 0u;
        __syncthreads();
        scan[b] += v;
        __syncthreads();
    }
    const u32 base = scan[b] - cnt;
    if (b < radix) {
        sg.first[b] = base;
        for (u32 k = 0; k < cnt && base + k < sg.max_segs; k++) sg.start[base + k] = lo + (u64)k * sg.seg_len;
    }
    if (b == radix - 1) {
        const u32 total = scan[b] < sg.max_segs ? scan[b] : sg.max_segs;
        sg.first[radix] = total;
        sg.start[total] = n;
        *sg.nseg = total;
    }
}

// rows[s][d] = keys of segment s whose digit is d
__global__ __launch_bounds__(T1_BLOCK) void tagpass_count_kernel(const u16* __restrict__ dig, TagSegs sg, u32* __restrict__ rows) {
    __shared__ u32 bins[T1_RADIX];
    const u32 s = blockIdx.x;
    if (s >= *sg.nseg) return;
    const int tid = threadIdx.x;
    bins[tid] = 0;
    __syncthreads();
    const u64 lo = sg.start[s], hi = sg.start[s + 1];
    // eight digits (16 bytes) a load, on the 16-byte grid of the array; the ends are cut by index
    const u64 a0 = lo & ~7ull;
    for (u64 base = a0 + 8ull * tid; base < hi; base += 8ull * T1_BLOCK) {
        const uint4 q = *reinterpret_cast<const uint4*>(dig + base);
        const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 g = base + i;
            const u32 d = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            if (g >= lo && g < hi) atomicAdd(&bins[d & (T1_RADIX - 1)], 1u);
        }
    }
    __syncthreads();
    rows[(u64)s * T1_RADIX + tid] = bins[tid];
}

// offs[s][d] = (keys with a smaller digit) + (keys of digit d in the segments before s); offs[nseg][d] = the end of digit d's bucket;
// cuts[d * radix0 + b] = where block (d, b) begins = offs[first segment of bucket b][d]; cuts[radix * radix0] = n
__global__ __launch_bounds__(T1_RADIX) void tagpass_scan_kernel(const u32* __restrict__ rows, TagSegs sg, u32 radix0, u64 n, u64* __restrict__ offs,
                                                                u64* __restrict__ cuts) {
    __shared__ u64 tot[T1_RADIX];
    const u32 d = threadIdx.x;
    const u32 nseg = *sg.nseg;
    u64 t = 0;
    for (u32 s = 0; s < nseg; s++) t += rows[(u64)s * T1_RADIX + d];
    tot[d] = t;
    __syncthreads();
    for (u32 o = 1; o < (u32)T1_RADIX; o <<= 1) {
        const u64 v = d >= o ? tot[d - o] : 0ull;
        __syncthreads();
        tot[d] += v;
        __syncthreads();
    }
    u64 run = tot[d] - t;
    for (u32 s = 0; s < nseg; s++) {
        offs[(u64)s * T1_RADIX + d] = run;
        run += rows[(u64)s * T1_RADIX + d];
    }
    offs[(u64)nseg * T1_RADIX + d] = run;
    __syncthreads();          // (this thread's own column only: the loop below reads what it wrote itself)
    for (u32 b = 0; b < radix0; b++) cuts[(u64)d * radix0 + b] = offs[(u64)sg.first[b] * T1_RADIX + d];
    if (d == 0) cuts[(u64)T1_RADIX * radix0] = n;
}

struct T1Args {
    const u32* tin;       // the keys' low words, ordered by the digit of the pass before ...
    const u16* din;       // ... and this pass's digit of each
    u64 n;
    TagSegs sg;
    const u64* offs;      // [nseg + 1][RADIX]
    const u32* rows;      // [nseg][RADIX]: checked against what the pass wrote
    u32* tout;
    u32* err;
};

// A unit of the output: up to G tags that lie side by side in exch and go to G consecutive, G-aligned places of the output (the
// first unit of a digit's piece may start off the grid, the last may be short).  One word: slot | (tags - 1) << 15 | digit << 19.
__device__ __forceinline__ u32 t1_unit_pack(u32 slot, u32 len, u32 digit) { return slot | ((len - 1u) << 15) | (digit << 19); }

__device__ __forceinline__ u32 t1_scan_dpp(u32 v) {          // inclusive prefix sum over the 64 lanes (row shifts, then the two row broadcasts)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

template <int NW>          // tags a thread takes per tile
struct T1Smem {
    static constexpr int TILE = T1_BLOCK * NW, CAP = TILE + T1_RADIX * (T1_G - 1), UNITS = CAP / T1_G + 2 * T1_RADIX;
    static_assert(CAP < (1 << 15), "t1_unit_pack");
    u32 exch[CAP + 64];            // the tile's tags, grouped by digit; what is left of a digit stays here until the next tile (+ a slot per lane for the dead)
    u64 gbase[T1_RADIX];           // output index of exch slot 0 as seen by this digit
    u32 units[UNITS];
    alignas(16) u32 cnt[T1_RADIX + 64];        // tags of the digit: the left-over ones between tiles, all of them after the ranking (+ 64 for the dead)
    u16 off[T1_RADIX];             // where the digit's tags start in exch
    alignas(16) u16 nu[T1_RADIX];              // units of the digit
    u32 nunits;
    u32 anybad;
};

// One digit per thread (RADIX == BLOCK): the digit's output cursor and its left-over tags' place are that thread's registers.
template <int NW>
__global__ __launch_bounds__(T1_BLOCK, 4) void tagpass_scatter_kernel(T1Args a) {
    using S = T1Smem<NW>;
    constexpr int RADIX = T1_RADIX, BLOCK = T1_BLOCK, G = T1_G, TILE = S::TILE;
    static_assert(RADIX == BLOCK && NW % 8 == 0, "one digit per thread; eight elements a load");
    __shared__ S sm;
    const u32 s = blockIdx.x;
    if (s >= *a.sg.nseg) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 lo = a.sg.start[s], hi = a.sg.start[s + 1];
    const u64 a0 = lo & ~7ull;          // tiles sit on the 8-element grid of the arrays: 16- and 32-byte loads, the segment's ends cut by index
    const u32 ntile = (u32)((hi - a0 + TILE - 1) / TILE);
    u64 F = a.offs[(u64)s * RADIX + tid];          // this thread's digit: its next output index ...
    const u64 Fend = F + a.rows[(u64)s * RADIX + tid];          // ... where its piece ends
    u32 nck = 0, ctail = 0;          // tags left over from the last tile, where they sit in exch
    u32 bad = 0;
    sm.cnt[tid] = 0;
    if (tid == 0) sm.anybad = 0;
    // this thread's elements of a tile: NW consecutive ones, eight to a load
    // The elements of tile t + 1 are taken out of the load registers, and those of tile t + 2 asked for, BEFORE the stores of tile t are
    // issued: a wait for loaded elements then never has a tile's stores in front of it (the memory counter is one for loads and stores).
    uint4 qd[NW / 8], qt[NW / 4], cd[NW / 8], ct[NW / 4];
    auto fetch = [&](u32 t) {
        const u64 base = a0 + (u64)t * TILE + (u64)NW * tid;
#pragma unroll
        for (int r = 0; r < NW / 8; r++) {
            const bool in = base + 8 * r < hi;          // (a load that begins inside the segment ends inside the arrays: they are padded)
            qd[r] = in ? *reinterpret_cast<const uint4*>(a.din + base + 8 * r) : make_uint4(0, 0, 0, 0);
            qt[2 * r] = in ? *reinterpret_cast<const uint4*>(a.tin + base + 8 * r) : make_uint4(0, 0, 0, 0);
            qt[2 * r + 1] = in ? *reinterpret_cast<const uint4*>(a.tin + base + 8 * r + 4) : make_uint4(0, 0, 0, 0);
        }
    };
    auto take = [&]() {          // the loaded elements become the current tile's (a use of every register: the wait for them is HERE)
#pragma unroll
        for (int r = 0; r < NW / 8; r++) { cd[r] = qd[r]; asm volatile("" : "+v"(cd[r].x), "+v"(cd[r].y), "+v"(cd[r].z), "+v"(cd[r].w)); }
#pragma unroll
        for (int r = 0; r < NW / 4; r++) { ct[r] = qt[r]; asm volatile("" : "+v"(ct[r].x), "+v"(ct[r].y), "+v"(ct[r].z), "+v"(ct[r].w)); }
    };
    fetch(0);
    take();
    if (ntile > 1) fetch(1);
    __syncthreads();
    // Units [first, end) of the list out: four lanes per unit, four tags (16 bytes) per lane, NF units of a lane group in flight
    struct __attribute__((packed, aligned(4))) Tag4 { u32 a, b, c, d; };
    auto store_units = [&](u32 first, u32 end) {
        constexpr int NF = 4;
        constexpr u32 LPU = G / 4, GROUPS = BLOCK / LPU;
        const u32 j = 4u * ((u32)tid & (LPU - 1)), g0 = (u32)tid / LPU;
        for (u32 base = first; base < end; base += NF * GROUPS) {
            u32 e4[NF];
            u32 k4[NF][4];
            u64 b4[NF];
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * GROUPS + g0;
                e4[g] = sm.units[u < (u32)S::UNITS ? u : 0u];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(e4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 sl = (e4[g] & 0x7fffu) + j;
#pragma unroll
                for (int q = 0; q < 4; q++) k4[g][q] = sm.exch[sl + q];          // (sl + q < CAP + 64: slots past a unit's end are read, never stored)
                b4[g] = sm.gbase[(e4[g] >> 19) & (u32)(RADIX - 1)];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(k4[g][0]), "+v"(k4[g][1]), "+v"(k4[g][2]), "+v"(k4[g][3]), "+v"(b4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * GROUPS + g0;
                const u32 len = ((e4[g] >> 15) & 15u) + 1u;
                u32* dst = a.tout + (b4[g] + (e4[g] & 0x7fffu) + j);          // (inside the digit's piece by construction)
                if (u < end) {
                    if (j + 4 <= len) { Tag4 v; v.a = k4[g][0]; v.b = k4[g][1]; v.c = k4[g][2]; v.d = k4[g][3]; *reinterpret_cast<Tag4*>(dst) = v; }
                    else {
#pragma unroll
                        for (int q = 0; q < 3; q++) if (j + q < len) dst[q] = k4[g][q];
                    }
                }
            }
        }
    };
    for (u32 t = 0; t < ntile; t++) {
        const bool last = t + 1 == ntile;
        // ---- this thread's digits and tags ------------------------------------------------------------------
        u32 dg[NW], tg[NW];
        u32 live = 0;
        {
            const u64 base = a0 + (u64)t * TILE + (u64)NW * tid;
#pragma unroll
            for (int r = 0; r < NW / 8; r++) {
                const u32 w[4] = {cd[r].x, cd[r].y, cd[r].z, cd[r].w};
                const u32 x[8] = {ct[2 * r].x, ct[2 * r].y, ct[2 * r].z, ct[2 * r].w, ct[2 * r + 1].x, ct[2 * r + 1].y, ct[2 * r + 1].z, ct[2 * r + 1].w};
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    dg[8 * r + i] = (w[i >> 1] >> (16 * (i & 1))) & (u32)(RADIX - 1);
                    tg[8 * r + i] = x[i];
                    const u64 g = base + 8 * r + i;
                    live |= (g >= lo && g < hi ? 1u : 0u) << (8 * r + i);
                }
            }
        }
        // ---- what the last tile left of this thread's digit: into registers, the park below moves it ----------
        u32 ck[G - 1];
#pragma unroll
        for (int j = 0; j < G - 1; j++) ck[j] = sm.exch[ctail + j];          // ctail + j < CAP + G: inside exch
        // ---- rank: the digit's counter hands out the places (a tile lies inside one bucket of the pass before: no order to keep) ----
        u32 rk[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) rk[i] = atomicAdd(&sm.cnt[((live >> i) & 1u) ? dg[i] : (u32)RADIX + (u32)lane], 1u);
        __syncthreads();
        // ---- per digit: how many tags leave now, where the digit sits in LDS ------------------------------------
        u32 O, tot, flen, head, nu;
        {
            const uint4 c0 = reinterpret_cast<const uint4*>(sm.cnt)[2 * lane], c1 = reinterpret_cast<const uint4*>(sm.cnt)[2 * lane + 1];
            const u32 s8 = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
            const u32 inc8 = t1_scan_dpp(s8);
            const u32 before = wave ? (u32)__builtin_amdgcn_readlane((int)inc8, 8 * wave - 1) : 0u;          // digits below 64 * wave
            tot = sm.cnt[tid];
            const u32 inc = t1_scan_dpp(tot);
            O = before + inc - tot;
            u64 end = F + tot;
            if (end > Fend) { bad = 1; sm.anybad = 1; end = Fend > F ? Fend : F; }          // never beyond the digit's piece (then the error word is set)
            u64 Eo = last ? end : (end & ~(u64)(G - 1));
            if (Eo < F) Eo = F;
            flen = (u32)(Eo - F);
            head = (u32)(F & (G - 1)) ? G - (u32)(F & (G - 1)) : 0u;
            if (head > flen) head = flen;
            nu = (head ? 1u : 0u) + (flen - head + G - 1) / G;
            sm.nu[tid] = (u16)nu;
        }
        __syncthreads();          // every wave has read the counters; the unit counts are there
        {
            const uint4 c0 = reinterpret_cast<const uint4*>(sm.nu)[lane];          // eight 16-bit counts: digits 8 l .. 8 l + 7
            const u32 s2 = c0.x + c0.y + c0.z + c0.w;
            const u32 s8 = (s2 & 0xffffu) + (s2 >> 16);
            const u32 inc8 = t1_scan_dpp(s8);
            const u32 before = wave ? (u32)__builtin_amdgcn_readlane((int)inc8, 8 * wave - 1) : 0u;
            const u32 inc = t1_scan_dpp(nu);
            u32 U = before + inc - nu;
            if (tid == 0) sm.nunits = (u32)__builtin_amdgcn_readlane((int)inc8, 63);
            sm.off[tid] = (u16)O;
            sm.cnt[tid] = tot - flen;
            sm.gbase[tid] = F - O;
            u32 slot = O, left = flen;
            if (head) { if (U < (u32)S::UNITS) sm.units[U] = t1_unit_pack(slot, head, (u32)tid); U++; slot += head; left -= head; }
            while (left) {
                const u32 len = left < (u32)G ? left : (u32)G;
                if (U < (u32)S::UNITS) sm.units[U] = t1_unit_pack(slot, len, (u32)tid);
                U++; slot += len; left -= len;
            }
            F += flen;
        }
        __syncthreads();
        // ---- park ------------------------------------------------------------------------------------------------
        {
            u32 at[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) at[i] = sm.off[dg[i]];
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const u32 slot = at[i] + rk[i];
                sm.exch[(((live >> i) & 1u) && slot < (u32)S::CAP) ? slot : (u32)S::CAP + (u32)lane] = tg[i];
            }
#pragma unroll
            for (int j = 0; j < G - 1; j++) sm.exch[((u32)j < nck && O + j < (u32)S::CAP) ? O + j : (u32)S::CAP + (u32)lane] = ck[j];
            nck = tot - flen;
            ctail = O + flen;
        }
        __syncthreads();
        if (!last) {
            take();          // tile t + 1 (asked for a tile ago)
            if (t + 2 < ntile) fetch(t + 2);
        }
        // ---- whole units out ---------------------------------------------------------------------------------------
        {
            const u32 nunits = (sm.anybad || sm.nunits > (u32)S::UNITS) ? 0u : sm.nunits;
            store_units(0, nunits);
        }
        __syncthreads();          // nobody still reads what the next tile's scan and park rewrite
    }
    if (bad || F != Fend || nck != 0) atomicOr(a.err, ZK_DERR_MISMATCH);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// tags / dig: pass 0's two arrays (u32[n], u16[n], ordered by digit d0 whose exclusive prefix is ghist0[0 .. radix0)), readable 16 bytes
// beyond n.  tout: u32[n].  cuts: [T1_RADIX * radix0 + 1] where the blocks (d1, d0) begin in tout.
int stream_pass1(zk_ctx* c, const u32* tags, const u16* dig, uint64_t n, const u64* ghist0, uint32_t radix0, int bits1, u32* tout, u64* cuts) {
    if (bits1 != T1_RBITS || radix0 > (uint32_t)T1_RADIX || radix0 == 0) return fail(c, ZK_EINTERNAL, "stream_pass1: digits of %d bits after %u buckets", bits1, radix0);
    if (((uintptr_t)tags & 15) || ((uintptr_t)dig & 15)) return fail(c, ZK_EINTERNAL, "stream_pass1: arrays off the 16-byte grid");
    TagSegs sg = {};
    // about eight segments per workgroup slot (two slots per CU), never shorter than a few tiles
    const uint64_t slots = 2ull * (uint64_t)(c->num_cus > 0 ? c->num_cus : 1);
    sg.seg_len = div_up(n, 8 * slots);
    if (sg.seg_len < 16384) sg.seg_len = 16384;
    sg.seg_len = (sg.seg_len + 7) & ~7ull;
    sg.max_segs = (u32)(n / sg.seg_len + radix0 + 1);
    u32* rows; u64* offs;
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)sg.max_segs + 1), (void**)&sg.start));
    ZK_TRY(arena_alloc(c, sizeof(u32) * ((uint64_t)T1_RADIX + 1), (void**)&sg.first));
    ZK_TRY(arena_alloc(c, sizeof(u32) * (uint64_t)sg.max_segs * T1_RADIX, (void**)&rows));
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)sg.max_segs + 1) * T1_RADIX, (void**)&offs));
    sg.nseg = (u32*)(c->d_scalars + 33);
    hipLaunchKernelGGL(tagpass_plan_kernel, dim3(1), dim3(T1_RADIX), 0, c->stream, ghist0, (u64)n, radix0, sg);
    prof_begin(c, ZK_PROF_HIST_ARRAY, 2 * n);
    hipLaunchKernelGGL(tagpass_count_kernel, dim3(sg.max_segs), dim3(T1_BLOCK), 0, c->stream, dig, sg, rows);
    prof_end(c);
    hipLaunchKernelGGL(tagpass_scan_kernel, dim3(1), dim3(T1_RADIX), 0, c->stream, (const u32*)rows, sg, radix0, (u64)n, offs, cuts);
    T1Args a = {};
    a.tin = tags; a.din = dig; a.n = n; a.sg = sg; a.offs = offs; a.rows = rows; a.tout = tout; a.err = c->d_err;
    prof_begin(c, ZK_PROF_PASS_KEYS, (2 + 4 + 4) * n);
    hipLaunchKernelGGL((tagpass_scatter_kernel<8>), dim3(sg.max_segs), dim3(T1_BLOCK), 0, c->stream, a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
