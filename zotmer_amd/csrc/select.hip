// select.hip -- order-preserving stream compaction kernels built on wave64 ballots:
//   K4  run-length count of a sorted k-mer array            (kmerize.merge, commands/kmerize.py:41-132)
//   K7  histogram of the counts                             (commands/kmerize.py:543-545, merge.py:88-92)
//   K8  prefix projection + adjacent dedupe                 (Measure.prep, commands/dist.py:43-49)
//   K10 count filter                                        (trim.trim, commands/trim.py:54-62)
//   K2  murmer subsample                                    (basics.sub, library/basics.py:252-259;
//                                                            caller commands/kmerize.py:494-509)
//   K1  encode to a list in reference order                 (basics.kmersList, library/basics.py:303-347)
//
// Common shape: a workgroup owns a tile of BLOCK*ITEMS consecutive elements in wave-striped
// order (row i of wave w = 64 consecutive elements, one per lane).  The keep/head flag of a row
// is one __ballot; a lane's output slot is  tile base + wave base + popcounts of earlier rows
// + mbcnt of its own row -- no LDS scan over elements.  The tile base comes from a decoupled
// look-back over the tiles (common.hpp).  Kept lanes of a row write consecutive addresses.
//
// Algorithmic bytes: one read of the input, one write of what survives.
#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "encode_tile.hpp"

namespace zk {

constexpr int SEL_BLOCK = 256;
constexpr int SEL_ITEMS = 8;
constexpr int SEL_TILE = SEL_BLOCK * SEL_ITEMS;
constexpr int SEL_NW = SEL_BLOCK / 64;

struct SelState {
    u64* status;
    u32* ticket;
    u32 ticket_base;
    u32 epoch;
    u32* err;
    u64* d_total;   // total number of outputs (written by the last tile)
    u32 tiles;
};

template <int NW>
struct SelSmemT {
    u32 wtot[NW];
    u64 tile_excl;
    u32 ticket;
};
typedef SelSmemT<SEL_NW> SelSmem;

// Returns the global output offset of this WAVE's first kept element; *tile_total gets the
// tile's number of kept elements.  All threads of the workgroup must call it.
template <int NW>
__device__ __forceinline__ u64 select_wave_base(SelSmemT<NW>& sm, const SelState& st, u32 tile, u32 wave_total, u32* tile_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm.wtot[wave] = wave_total;
    __syncthreads();
    u32 wex = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w < wave) wex += sm.wtot[w];
        tot += sm.wtot[w];
    }
    if (wave == 0) {
        u64 ex = lookback_exclusive(st.status, tile, tot, st.epoch, st.err);
        if (lane == 0) {
            sm.tile_excl = ex;
            if (tile == st.tiles - 1) *st.d_total = ex + tot;
        }
    }
    __syncthreads();
    *tile_total = tot;
    return sm.tile_excl + wex;
}

// ---------------------------------------------------------------------------------------
// K4: run-length count.  uniq[j], counts[j] for the j-th distinct key of a sorted array.
// A run that crosses a tile boundary is finished by rle_fixup_kernel: every tile records how
// many of its leading elements continue the previous tile's last run.
// ---------------------------------------------------------------------------------------
constexpr int RLE_BLOCK = 512;
constexpr int RLE_ITEMS = 16;
constexpr int RLE_TILE = RLE_BLOCK * RLE_ITEMS;
constexpr int RLE_NW = RLE_BLOCK / 64;
constexpr u32 RLE_NONE = 0xFFFFFFFFu;

struct RleSmem {
    SelSmemT<RLE_NW> sel;
    u32 wfirst[RLE_NW];     // tile-relative position of each wave's first head, or RLE_NONE
};

// `uniq` may alias `keys` (in-place): a tile learns its output offset only after every earlier
// tile has loaded its inputs, outputs never land beyond the tile's own input range, and the one
// neighbour element a later tile may still read can only be overwritten with its own value.
//
// Everything stays in registers: the head flags of a row are one ballot, a head's run length is
// the distance to the next set bit (same row, a later row of the wave, a later wave via LDS, or
// the end of the tile), its output slot a popcount.  8192 keys per tile keep the tile rate -- and
// with it the length of the look-back chains -- low.
// pack > 0: uniq[j] holds (key << pack) | min(count, 2^pack - 1) -- key and count in ONE word, so that the passes which
// finish the sort move 8 bytes per entry through the fast key kernel instead of 12 through the pair kernel.  Should a
// count not fit, *ovf is set and the caller starts over without packing, so in this mode uniq must NOT alias keys (the
// in-place form also relies on an overwritten neighbour keeping its value, which a packed word does not); counts may be
// null.
__global__ __launch_bounds__(RLE_BLOCK) void rle_kernel(const u64* keys, u64 n, u64* uniq,
                                                        u32* __restrict__ counts, u64 cap, u32* __restrict__ lead,
                                                        SelState st, int pack, u32* ovf) {
    __shared__ RleSmem sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.sel.ticket) - st.ticket_base;
    const u64 tile_base = (u64)tile * RLE_TILE;
    const u32 wave_rel = (u32)wave * (64 * RLE_ITEMS);
    const u64 base = tile_base + wave_rel;

    u64 k[RLE_ITEMS];
    u64 hm[RLE_ITEMS];
    u32 wave_total = 0;
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        k[i] = (idx < n) ? keys[idx] : 0ull;
    }
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        u64 prev = __shfl_up(k[i], 1, 64);
        if (i > 0) {
            const u64 last = __shfl(k[i - 1], 63, 64);   // lane 63 of the previous row
            if (lane == 0) prev = last;
        } else if (lane == 0) {
            prev = (idx > 0 && idx < n) ? keys[idx - 1] : 0ull;   // element before the wave's segment
        }
        const bool head = (idx < n) && (idx == 0 || k[i] != prev);
        hm[i] = __ballot(head);
        wave_total += (u32)__popcll(hm[i]);
    }
    // wave-relative position of the first head after row i, and of the wave's first head
    u32 nh[RLE_ITEMS];
    u32 run = RLE_NONE;
#pragma unroll
    for (int i = RLE_ITEMS - 1; i >= 0; i--) {
        nh[i] = run;
        if (hm[i]) run = (u32)(i * 64) + (u32)__builtin_ctzll(hm[i]);
    }
    if (lane == 0) sm.wfirst[wave] = (run == RLE_NONE) ? RLE_NONE : wave_rel + run;

    u32 tile_total;
    const u64 wbase = select_wave_base(sm.sel, st, tile, wave_total, &tile_total);   // barriers inside
    const u64 rem = n - tile_base;
    const u32 tile_len = rem < (u64)RLE_TILE ? (u32)rem : (u32)RLE_TILE;
    u32 next_wave = tile_len;      // first head in a later wave of the tile, else the tile's end
#pragma unroll
    for (int w = RLE_NW - 1; w >= 0; w--)
        if (w > wave && sm.wfirst[w] != RLE_NONE) next_wave = sm.wfirst[w];

    u64 q = wbase;
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        if ((hm[i] >> lane) & 1ull) {
            const u64 above = (hm[i] >> lane) >> 1;
            const u32 me = wave_rel + (u32)(i * 64 + lane);
            u32 nxt;
            if (above) nxt = me + 1 + (u32)__builtin_ctzll(above);
            else nxt = (nh[i] != RLE_NONE) ? wave_rel + nh[i] : next_wave;
            const u64 pos = q + popc_below(hm[i]);
            if (pos < cap) {
                const u32 len = nxt - me;
                if (pack) {
                    const u32 top = (1u << pack) - 1u;
                    if (len > top) atomicOr(ovf, 1u);
                    uniq[pos] = (k[i] << pack) | (u64)(len < top ? len : top);
                } else {
                    uniq[pos] = k[i];
                }
                if (counts) counts[pos] = len;
            }
        }
        q += (u32)__popcll(hm[i]);
    }
    if (threadIdx.x == 0) {
        u32 first = tile_len;
#pragma unroll
        for (int w = RLE_NW - 1; w >= 0; w--)
            if (sm.wfirst[w] != RLE_NONE) first = sm.wfirst[w];
        lead[tile] = first;
        if (tile == st.tiles - 1 && sm.sel.tile_excl + tile_total > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
    }
}

// ---------------------------------------------------------------------------------------
// K4': run-length count of an array that is sorted by its top bits only (prefix = key >> pshift).
//
// The radix sort is stopped early: with T >= log2(n) + 3.5 sorted bits almost every group of equal
// prefix holds ONE distinct k-mer, so it is already where a full sort would put it.  A group is
// written to the sorted main list iff it lies inside one tile and is uniform (one distinct key);
// every other group -- mixed, or cut by a tile edge -- is appended, entry by entry, to an unordered
// side list that the caller sorts with the strand mirror (which it has to sort anyway).  Whatever
// the input, the two lists together hold every (k-mer, run length) exactly once; only how much lands
// in the side list depends on the data.
// ---------------------------------------------------------------------------------------
constexpr int PFX_WORDS = RLE_TILE / 64;

struct PfxSmem {
    SelSmemT<RLE_NW> sel;
    u64 fh[PFX_WORDS];      // full-key heads, one bit per tile position
    u64 ph[PFX_WORDS];      // prefix heads
    u8 cls[RLE_TILE];       // per group start: 1 = goes to the main list
    u32 wside[RLE_NW];
    u64 side_base;
};

__device__ __forceinline__ int bits_prev(const u64* b, int p) {          // largest set q <= p, or -1
    int w = p >> 6;
    u64 m = b[w] & (~0ull >> (63 - (p & 63)));
    while (true) {
        if (m) return (w << 6) + 63 - __builtin_clzll(m);
        if (--w < 0) return -1;
        m = b[w];
    }
}
__device__ __forceinline__ int bits_next(const u64* b, int p, int limit) {   // smallest set q > p below limit, or limit
    if (p + 1 >= limit) return limit;
    int w = (p + 1) >> 6;
    u64 m = b[w] & (~0ull << ((p + 1) & 63));
    const int lw = (limit - 1) >> 6;
    while (true) {
        if (m) { const int q = (w << 6) + __builtin_ctzll(m); return q < limit ? q : limit; }
        if (++w > lw) return limit;
        m = b[w];
    }
}

__global__ __launch_bounds__(RLE_BLOCK) void rle_prefix_kernel(const u64* keys, u64 n, int pshift, u64* uniq, u32* __restrict__ counts,
                                                               u64 cap, u64* __restrict__ side_k, u32* __restrict__ side_c, u64 side_cap,
                                                               u64* side_n, SelState st) {
    __shared__ PfxSmem sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.sel.ticket) - st.ticket_base;
    const u64 tile_base = (u64)tile * RLE_TILE;
    const u32 wave_rel = (u32)wave * (64 * RLE_ITEMS);
    const u64 base = tile_base + wave_rel;
    const u64 rem = n - tile_base;
    const int tile_len = rem < (u64)RLE_TILE ? (int)rem : RLE_TILE;

    u64 k[RLE_ITEMS];
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        k[i] = (idx < n) ? keys[idx] : 0ull;
    }
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        u64 prev = __shfl_up(k[i], 1, 64);
        if (i > 0) {
            const u64 last = __shfl(k[i - 1], 63, 64);
            if (lane == 0) prev = last;
        } else if (lane == 0) {
            prev = (idx > 0 && idx < n) ? keys[idx - 1] : 0ull;
        }
        const bool in = idx < n;
        const bool fhead = in && (idx == 0 || k[i] != prev);
        const bool phead = in && (idx == 0 || (k[i] >> pshift) != (prev >> pshift));
        const u64 fm = __ballot(fhead), pm = __ballot(phead);
        if (lane == 0) { sm.fh[wave * RLE_ITEMS + i] = fm; sm.ph[wave * RLE_ITEMS + i] = pm; }
    }
    // does the group that is open at the end of the tile continue in the next one?
    bool cont_right = false;
    if (threadIdx.x == 0) {
        const u64 nxt = tile_base + (u64)tile_len;
        if (nxt < n && tile_len > 0) cont_right = (keys[nxt] >> pshift) == (keys[nxt - 1] >> pshift);
        sm.wside[0] = cont_right ? 1u : 0u;      // parked here until the barrier below
    }
    __syncthreads();
    cont_right = sm.wside[0] != 0;
    __syncthreads();

    // ---- classify every group by its first element (a prefix head) -------------------------------
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const int p = (int)wave_rel + i * 64 + lane;
        if (p < tile_len && ((sm.ph[p >> 6] >> (p & 63)) & 1ull)) {
            const int ge = bits_next(sm.ph, p, tile_len);             // first position of the next group, or tile_len
            const bool open_right = (ge == tile_len) && cont_right;
            const bool mixed = bits_next(sm.fh, p, ge) < ge;            // another full head inside the group
            sm.cls[p] = (!open_right && !mixed) ? 1 : 0;
        }
    }
    __syncthreads();

    // ---- every full head is one (key, run length) entry: main or side ------------------------------
    u64 mm[RLE_ITEMS];           // main entries of the row
    u64 sd[RLE_ITEMS];           // side entries of the row
    u32 len[RLE_ITEMS];
    u32 wave_main = 0, wave_side = 0;
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const int p = (int)wave_rel + i * 64 + lane;
        bool is_main = false, is_side = false;
        len[i] = 0;
        if (p < tile_len && ((sm.fh[p >> 6] >> (p & 63)) & 1ull)) {
            const int nx = bits_next(sm.fh, p, tile_len);
            len[i] = (u32)(nx - p);
            const int gs = bits_prev(sm.ph, p);                        // start of my group, -1: it began in an earlier tile
            is_main = (gs >= 0) && sm.cls[gs];
            is_side = !is_main;
        }
        mm[i] = __ballot(is_main);
        sd[i] = __ballot(is_side);
        wave_main += (u32)__popcll(mm[i]);
        wave_side += (u32)__popcll(sd[i]);
    }
    // leading elements of the tile that continue the previous tile's last RUN (no full head of their
    // own) form one more side entry, owned by thread 0
    int lead_len = 0;
    if (threadIdx.x == 0 && tile_len > 0 && !(sm.fh[0] & 1ull)) lead_len = bits_next(sm.fh, 0, tile_len);
    if (lane == 0) sm.wside[wave] = wave_side + (lead_len ? 1u : 0u);

    u32 tile_main;
    const u64 wbase = select_wave_base(sm.sel, st, tile, wave_main, &tile_main);     // barriers inside
    if (threadIdx.x == 0) {
        u32 tot = 0;
#pragma unroll
        for (int w = 0; w < RLE_NW; w++) tot += sm.wside[w];
        sm.side_base = tot ? atomicAdd(side_n, (u64)tot) : 0ull;
    }
    __syncthreads();
    u32 sex = 0;
#pragma unroll
    for (int w = 0; w < RLE_NW; w++) if (w < wave) sex += sm.wside[w];
    u64 qm = wbase, qs = sm.side_base + sex;
    if (lead_len) {                                   // thread 0 only; k[0] is the tile's first key
        if (qs < side_cap) { side_k[qs] = k[0]; side_c[qs] = (u32)lead_len; }
    }
    if (wave == 0) qs += (sm.wside[0] > wave_side) ? 1 : 0;     // wave 0's entries follow the lead entry
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        if ((mm[i] >> lane) & 1ull) {
            const u64 pos = qm + popc_below(mm[i]);
            if (pos < cap) { uniq[pos] = k[i]; counts[pos] = len[i]; }
        }
        if ((sd[i] >> lane) & 1ull) {
            const u64 pos = qs + popc_below(sd[i]);
            if (pos < side_cap) { side_k[pos] = k[i]; side_c[pos] = len[i]; }
        }
        qm += (u32)__popcll(mm[i]);
        qs += (u32)__popcll(sd[i]);
    }
    if (threadIdx.x == 0 && tile == st.tiles - 1 && sm.sel.tile_excl + tile_main > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
}

int rle_prefix(zk_ctx* c, const u64* sorted, uint64_t n, int pshift, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_main,
               u64* side_k, u32* side_c, uint64_t side_cap, uint64_t* n_side) {
    *n_main = 0; *n_side = 0;
    if (n == 0) return ZK_OK;
    SelState st;
    st.tiles = (u32)div_up(n, RLE_TILE);
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    u64* d_side = c->d_scalars + 10;
    ZK_HIP(c, hipMemsetAsync(d_side, 0, sizeof(u64), c->stream));
    prof_begin(c, ZK_PROF_RLE, 8 * n);
    hipLaunchKernelGGL(rle_prefix_kernel, dim3(st.tiles), dim3(RLE_BLOCK), 0, c->stream, sorted, (u64)n, pshift, uniq, counts, (u64)cap,
                       side_k, side_c, (u64)side_cap, d_side, st);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, 2 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_main = c->h_scalars[9];
    *n_side = c->h_scalars[10];
    return check_device_error(c);
}

// ---------------------------------------------------------------------------------------
// sum the payloads of equal adjacent keys: (sorted keys with duplicates, w) -> (distinct keys, sums).
// Element i belongs to run number (heads up to and including i) - 1.  Inside a tile the weights of a run meet in one LDS
// word (LDS atomics; runs are short: the pieces of a run that the early collapse split, or a k-mer and its mirror), every
// run that starts in the tile is then written once with plain stores; the leading elements of a tile that continue the
// previous tile's last run leave their sum in lead[tile] and a second launch adds it where it belongs (as rle_fixup does).
// ---------------------------------------------------------------------------------------
struct RbkSmem {
    SelSmemT<RLE_NW> sel;
    u32 sum[RLE_TILE];        // per run that starts in this tile: sum of its weights inside the tile
    u32 lead;                 // sum of the weights of the leading elements that belong to the previous tile's run
};

// pack > 0: keys[] holds (key << pack) | weight and w is not read; maxsum (may be null) gets the largest sum written
__global__ __launch_bounds__(RLE_BLOCK) void reduce_by_key_kernel(const u64* __restrict__ keys, const u32* __restrict__ w, u64 n,
                                                                  u64* __restrict__ uniq, u32* __restrict__ sums, u64 cap,
                                                                  u32* __restrict__ lead, SelState st, int pack, u32* maxsum) {
    __shared__ RbkSmem sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.sel.ticket) - st.ticket_base;
    const u64 base = (u64)tile * RLE_TILE + (u64)wave * (64 * RLE_ITEMS);
    for (int i = threadIdx.x; i < RLE_TILE; i += RLE_BLOCK) sm.sum[i] = 0;
    if (threadIdx.x == 0) sm.lead = 0;
    u64 k[RLE_ITEMS];
    u32 wt[RLE_ITEMS];
    u64 hm[RLE_ITEMS];
    u32 wave_total = 0;
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        k[i] = (idx < n) ? keys[idx] : 0ull;
        if (pack) {
            wt[i] = (u32)(k[i] & ((1ull << pack) - 1ull));
            k[i] >>= pack;
        } else {
            wt[i] = (idx < n) ? w[idx] : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        u64 prev = __shfl_up(k[i], 1, 64);
        if (i > 0) {
            const u64 last = __shfl(k[i - 1], 63, 64);
            if (lane == 0) prev = last;
        } else if (lane == 0) {
            prev = (idx > 0 && idx < n) ? (keys[idx - 1] >> pack) : 0ull;
        }
        const bool head = (idx < n) && (idx == 0 || k[i] != prev);
        hm[i] = __ballot(head);
        wave_total += (u32)__popcll(hm[i]);
    }
    u32 tile_total;
    const u64 wbase = select_wave_base(sm.sel, st, tile, wave_total, &tile_total);      // barriers inside: sm.sum is zeroed
    const u64 tile_first = sm.sel.tile_excl;                  // output slot of the first run that starts in this tile
    u32 q = (u32)(wbase - tile_first);                         // runs started in the tile before this wave's first row
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        if (idx < n) {
            const u32 incl = q + popc_below(hm[i]) + (u32)((hm[i] >> lane) & 1ull);     // heads of the tile up to and including me
            u32 old;
            if (incl == 0) old = atomicAdd(&sm.lead, wt[i]);
            else old = atomicAdd(&sm.sum[incl - 1], wt[i]);
            if (old + wt[i] < old) atomicOr(st.err, ZK_DERR_COUNT_OVERFLOW);
        }
        q += (u32)__popcll(hm[i]);
    }
    __syncthreads();
    q = (u32)(wbase - tile_first);
    u32 mx = 0;
#pragma unroll
    for (int i = 0; i < RLE_ITEMS; i++) {
        if ((hm[i] >> lane) & 1ull) {
            const u32 r = q + popc_below(hm[i]);
            const u64 pos = tile_first + r;
            if (pos < cap) { uniq[pos] = k[i]; sums[pos] = sm.sum[r]; }
            if (sm.sum[r] > mx) mx = sm.sum[r];
        }
        q += (u32)__popcll(hm[i]);
    }
    if (maxsum) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
        // one word takes ~88 atomics per microsecond: only a wave that beats the value it can see asks for the atomic
        if (lane == 0 && mx > __hip_atomic_load(maxsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxsum, mx);
    }
    if (threadIdx.x == 0) {
        lead[tile] = sm.lead;
        if (tile == st.tiles - 1 && tile_first + tile_total > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
    }
}

// sums[(first output slot of tile t) - 1] += lead[t]
__global__ void reduce_fixup_kernel(const u32* __restrict__ lead, const u64* __restrict__ status, u32 tiles, u32* sums, u64 cap, u32* err,
                                    u32* maxsum) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 || t >= tiles) return;
    const u32 l = lead[t];
    if (l == 0) return;
    const u64 excl = status[t - 1] & ZK_ST_VALUE_MASK;   // inclusive prefix of the tile before = first slot of this tile
    if (excl == 0 || excl - 1 >= cap) return;
    const u32 old = atomicAdd(&sums[excl - 1], l);
    if (old + l < old) atomicOr(err, ZK_DERR_COUNT_OVERFLOW);
    if (maxsum && old + l > __hip_atomic_load(maxsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxsum, old + l);
}

// pack > 0: `sorted` holds (key << pack) | weight, `w` is ignored.  max_sum (may be null): the largest sum written.
int reduce_by_key(zk_ctx* c, const u64* sorted, const u32* w, uint64_t n, u64* uniq, u32* sums, uint64_t cap, uint64_t* n_out, int pack,
                  uint64_t* max_sum) {
    *n_out = 0;
    if (max_sum) *max_sum = 0;
    if (n == 0) return ZK_OK;
    u32* d_max = max_sum ? (u32*)(c->d_scalars + 25) : nullptr;
    if (d_max) ZK_HIP(c, hipMemsetAsync(d_max, 0, sizeof(u64), c->stream));
    SelState st;
    st.tiles = (u32)div_up(n, RLE_TILE);
    u32* lead;
    ZK_TRY(arena_alloc(c, sizeof(u32) * st.tiles, (void**)&lead));
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    prof_begin(c, ZK_PROF_SELECT, (pack ? 8 : 12) * n);
    hipLaunchKernelGGL(reduce_by_key_kernel, dim3(st.tiles), dim3(RLE_BLOCK), 0, c->stream, sorted, w, (u64)n, uniq, sums, (u64)cap, lead, st,
                       pack, d_max);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(reduce_fixup_kernel, dim3((u32)div_up(st.tiles, 256)), dim3(256), 0, c->stream, lead, c->status, st.tiles, sums,
                       (u64)cap, c->d_err, d_max);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (d_max) ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 25, c->d_scalars + 25, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = c->h_scalars[9];
    if (max_sum) *max_sum = c->h_scalars[25] & 0xffffffffull;
    return check_device_error(c);
}

// how many elements of keys[0, m) differ from their predecessor (m = min(n, 2^18)): the yield a run-length collapse of the
// array would have, estimated from its head -- after LSD passes over the low bits the head of the array is an unbiased sample
// of the distinct k-mers with all their copies
__global__ void sample_heads_kernel(const u64* __restrict__ keys, u64 m, u64* out) {
    u32 h = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x)
        h += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    h = wave_sum_u32(h);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, (u64)h);
}

int sample_heads(zk_ctx* c, const u64* keys, uint64_t n, uint64_t* sampled, uint64_t* heads) {
    const uint64_t m = n < (1ull << 18) ? n : (1ull << 18);
    *sampled = m; *heads = m;
    if (m == 0) return ZK_OK;
    u64* d = c->d_scalars + 23;
    ZK_HIP(c, hipMemsetAsync(d, 0, sizeof(u64), c->stream));
    hipLaunchKernelGGL(sample_heads_kernel, dim3(64), dim3(256), 0, c->stream, keys, (u64)m, d);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 23, d, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *heads = c->h_scalars[23];
    return ZK_OK;
}

// counts[(first output index of tile t) - 1] += lead[t]
__global__ void rle_fixup_kernel(const u32* __restrict__ lead, const u64* __restrict__ status, u32 tiles, u32* counts,
                                 u64 cap, u32* err, u64* packed, int pack, u32* ovf) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 || t >= tiles) return;
    const u32 l = lead[t];
    if (l == 0) return;
    const u64 excl = status[t - 1] & ZK_ST_VALUE_MASK;   // inclusive prefix of the tile before
    if (excl == 0 || excl - 1 >= cap) return;
    if (counts) {
        const u32 old = atomicAdd(&counts[excl - 1], l);
        if (old + l < old) atomicOr(err, ZK_DERR_COUNT_OVERFLOW);
    }
    if (pack) {
        // the count field of the packed word: add without ever carrying into the key (several tiles may add to one run)
        const u64 top = (1ull << pack) - 1ull;
        u64 w = packed[excl - 1];
        while (true) {
            const u64 cnt = (w & top) + l;
            if (cnt > top) atomicOr(ovf, 1u);
            const u64 nw = (w & ~top) | (cnt < top ? cnt : top);
            const u64 seen = atomicCAS((unsigned long long*)&packed[excl - 1], (unsigned long long)w, (unsigned long long)nw);
            if (seen == w) break;
            w = seen;
        }
    }
}

int rle(zk_ctx* c, const u64* sorted, uint64_t n, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_unique, int pack, bool* overflow) {
    *n_unique = 0;
    if (overflow) *overflow = false;
    if (n == 0) return ZK_OK;
    u32* d_ovf = (u32*)(c->d_scalars + 24);
    if (pack) ZK_HIP(c, hipMemsetAsync(d_ovf, 0, sizeof(u64), c->stream));
    SelState st;
    st.tiles = (u32)div_up(n, RLE_TILE);
    u32* lead;
    ZK_TRY(arena_alloc(c, sizeof(u32) * st.tiles, (void**)&lead));
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    prof_begin(c, ZK_PROF_RLE, 8 * n);
    hipLaunchKernelGGL(rle_kernel, dim3(st.tiles), dim3(RLE_BLOCK), 0, c->stream, sorted, (u64)n, uniq, counts, (u64)cap, lead, st, pack, d_ovf);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(rle_fixup_kernel, dim3((u32)div_up(st.tiles, 256)), dim3(256), 0, c->stream, lead, c->status, st.tiles,
                       counts, (u64)cap, c->d_err, uniq, pack, d_ovf);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    if (pack) ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 24, c->d_scalars + 24, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_unique = c->h_scalars[9];
    if (pack && overflow) *overflow = (c->h_scalars[24] & 0xffffffffull) != 0;
    ZK_TRY(check_device_error(c));
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
// generic flag -> compact kernels
// ---------------------------------------------------------------------------------------
struct TrimOp {          // K10
    const u64* keys; const void* cnts; int cbits; u64 lo, hi; u64* ok; void* oc;
    struct R { u64 k; u64 c; };
    __device__ bool load(u64 idx, R& r) const {
        r.k = keys[idx];
        r.c = (cbits == 32) ? (u64)((const u32*)cnts)[idx] : ((const u64*)cnts)[idx];
        return r.c >= lo && (hi == 0 || r.c <= hi);
    }
    __device__ void store(u64 pos, const R& r) const {
        ok[pos] = r.k;
        if (cbits == 32) ((u32*)oc)[pos] = (u32)r.c; else ((u64*)oc)[pos] = r.c;
    }
};
struct DedupeOp {        // K8
    const u64* keys; int shift; u64* out;
    struct R { u64 y; };
    __device__ bool load(u64 idx, R& r) const {
        r.y = keys[idx] >> shift;
        return idx == 0 || (keys[idx - 1] >> shift) != r.y;
    }
    __device__ void store(u64 pos, const R& r) const { out[pos] = r.y; }
};
struct SubsampleOp {     // K2: float(murmer(x, seed)) / float(2**61 - 1) < p, in doubles as the reference
    const u64* keys; u64 seed; double p; u64* out;
    struct R { u64 k; };
    __device__ bool load(u64 idx, R& r) const {
        r.k = keys[idx];
        const double u = (double)murmer(r.k, seed) / (double)0x1FFFFFFFFFFFFFFFull;
        return u < p;
    }
    __device__ void store(u64 pos, const R& r) const { out[pos] = r.k; }
};

struct SubPairOp {       // K2 applied to an already counted set, in place (loads precede any store
                         // that could alias them: a tile learns its offset only after every
                         // earlier tile has loaded its inputs)
    u64* keys; u32* cnts; u64 seed; double p;
    struct R { u64 k; u32 c; };
    __device__ bool load(u64 idx, R& r) const {
        r.k = keys[idx]; r.c = cnts[idx];
        const double u = (double)murmer(r.k, seed) / (double)0x1FFFFFFFFFFFFFFFull;
        return u < p;
    }
    __device__ void store(u64 pos, const R& r) const { keys[pos] = r.k; cnts[pos] = r.c; }
};

struct SampleOp {        // sample.sampleD (zotmer/commands/sample.py:27-34): 40 hash bits as a fraction, in doubles
    const u64* keys; const u64* cnts; u64 seed; double p; u64* ok; u64* oc;
    struct R { u64 k; u64 c; };
    __device__ bool load(u64 idx, R& r) const {
        r.k = keys[idx]; r.c = cnts[idx];
        const u64 M = 0xFFFFFFFFFFull;
        const double u = (double)(murmer(r.k, seed) & M) / (double)M;
        return u < p;
    }
    __device__ void store(u64 pos, const R& r) const { ok[pos] = r.k; oc[pos] = r.c; }
};

template <class Op>
__global__ __launch_bounds__(SEL_BLOCK) void select_kernel(Op op, u64 n, u64 cap, SelState st) {
    __shared__ SelSmem sm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const u64 base = (u64)tile * SEL_TILE + (u64)wave * (64 * SEL_ITEMS);
    typename Op::R r[SEL_ITEMS];
    u64 km[SEL_ITEMS];
    u32 wave_total = 0;
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++) {
        const u64 idx = base + (u64)i * 64 + lane;
        const bool keep = (idx < n) && op.load(idx, r[i]);
        km[i] = __ballot(keep);
        wave_total += (u32)__popcll(km[i]);
    }
    u32 tile_total;
    u64 pos = select_wave_base(sm, st, tile, wave_total, &tile_total);
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++) {
        if ((km[i] >> lane) & 1ull) {
            const u64 q = pos + popc_below(km[i]);
            if (q < cap) op.store(q, r[i]);
        }
        pos += (u32)__popcll(km[i]);
    }
    if (threadIdx.x == 0 && tile == st.tiles - 1 && sm.tile_excl + tile_total > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
}

template <class Op>
static int run_select(zk_ctx* c, const Op& op, uint64_t n, uint64_t cap, uint64_t* n_out) {
    *n_out = 0;
    if (n == 0) return ZK_OK;
    SelState st;
    st.tiles = (u32)div_up(n, SEL_TILE);
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    prof_begin(c, ZK_PROF_SELECT, sizeof(typename Op::R) * n);
    hipLaunchKernelGGL((select_kernel<Op>), dim3(st.tiles), dim3(SEL_BLOCK), 0, c->stream, op, (u64)n, (u64)cap, st);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = c->h_scalars[9];
    return check_device_error(c);
}

int trim(zk_ctx* c, const u64* keys, const void* cnts, int cbits, uint64_t n, u64 lo, u64 hi, u64* ok, void* oc,
         uint64_t cap, uint64_t* n_out) {
    TrimOp op{keys, cnts, cbits, lo, hi, ok, oc};
    return run_select(c, op, n, cap, n_out);
}
int project_dedupe(zk_ctx* c, const u64* keys, uint64_t n, int shift, u64* out, uint64_t cap, uint64_t* n_out) {
    DedupeOp op{keys, shift, out};
    return run_select(c, op, n, cap, n_out);
}
int subsample_pairs(zk_ctx* c, u64* keys, u32* cnts, uint64_t n, u64 seed, double p, uint64_t* n_out) {
    SubPairOp op{keys, cnts, seed, p};
    return run_select(c, op, n, n, n_out);
}
int sample_pairs(zk_ctx* c, const u64* keys, const u64* cnts, uint64_t n, u64 seed, double p, u64* ok, u64* oc, uint64_t cap,
                 uint64_t* n_out) {
    SampleOp op{keys, cnts, seed, p, ok, oc};
    return run_select(c, op, n, cap, n_out);
}
int subsample(zk_ctx* c, const u64* keys, uint64_t n, u64 seed, double p, u64* out, uint64_t cap, uint64_t* n_out) {
    SubsampleOp op{keys, seed, p, out};
    return run_select(c, op, n, cap, n_out);
}

// ---------------------------------------------------------------------------------------
// K1 as a list: k-mers of a base stream in the reference's order (x then rc(x) per window)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(SEL_BLOCK) void encode_list_kernel(const u8* __restrict__ stream, u64 n_bytes, int K, int both,
                                                                u64* __restrict__ out, u64 cap, u64* acgt, SelState st) {
    __shared__ SelSmem sm;
    __shared__ TileImage<SEL_TILE> img;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    stage_tile<SEL_BLOCK, SEL_TILE>(stream, n_bytes, (u64)tile * SEL_TILE, img);
    u64 x[SEL_ITEMS];
    u64 km[SEL_ITEMS];
    u32 wave_total = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++) {
        const int p = wave * (64 * SEL_ITEMS) + i * 64 + lane;
        const bool ok = window_at(img, p, K, x[i]);
        km[i] = __ballot(ok);
        wave_total += (u32)__popcll(km[i]);
        if (ok) {
            const u32 f = (u32)(x[i] & 3), b = (u32)(revcomp(K, x[i]) & 3);
            a0 += (f == 0) + (both && b == 0); a1 += (f == 1) + (both && b == 1);
            a2 += (f == 2) + (both && b == 2); a3 += (f == 3) + (both && b == 3);
        }
    }
    u32 tile_total;
    u64 pos = select_wave_base(sm, st, tile, wave_total, &tile_total);
    const u64 mul = both ? 2 : 1;
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++) {
        if ((km[i] >> lane) & 1ull) {
            const u64 q = (pos + popc_below(km[i])) * mul;
            if (q + mul <= cap) {
                out[q] = x[i];
                if (both) out[q + 1] = revcomp(K, x[i]);
            }
        }
        pos += (u32)__popcll(km[i]);
    }
    if (threadIdx.x == 0 && tile == st.tiles - 1 && (sm.tile_excl + tile_total) * mul > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
    if (acgt) {      // one partial row per tile, summed afterwards (no contended atomics)
        __shared__ u64 scratch[SEL_NW];
        const u64 t0 = block_sum_u64(a0, scratch), t1 = block_sum_u64(a1, scratch), t2 = block_sum_u64(a2, scratch),
                  t3 = block_sum_u64(a3, scratch);
        if (threadIdx.x == 0) { u64* row = acgt + 4ull * tile; row[0] = t0; row[1] = t1; row[2] = t2; row[3] = t3; }
    }
}

int encode_list(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int both, u64* out, uint64_t cap, uint64_t* n_out,
                uint64_t acgt[4]) {
    *n_out = 0;
    if (acgt) acgt[0] = acgt[1] = acgt[2] = acgt[3] = 0;
    if (n_bytes == 0) return ZK_OK;
    if ((uintptr_t)stream & 15) return fail(c, ZK_EINVAL, "base stream must be 16-byte aligned");
    SelState st;
    st.tiles = (u32)div_up(n_bytes, SEL_TILE);
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    u64* d_rows;
    ZK_TRY(arena_alloc(c, 32ull * st.tiles, (void**)&d_rows));
    hipLaunchKernelGGL(encode_list_kernel, dim3(st.tiles), dim3(SEL_BLOCK), 0, c->stream, stream, (u64)n_bytes, K, both, out,
                       (u64)cap, d_rows, st);
    ZK_HIP(c, hipGetLastError());
    ZK_TRY(column_sum(c, d_rows, st.tiles, 4, c->d_scalars + 0));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(u64) * 16, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = c->h_scalars[9] * (both ? 2 : 1);
    if (acgt) for (int b = 0; b < 4; b++) acgt[b] = c->h_scalars[b];
    return check_device_error(c);
}

// ---------------------------------------------------------------------------------------
// K2b: capture mode, `zot kmerize -C BAITS` (commands/kmerize.py:480-485,510-520): a read
// contributes ALL its k-mers iff any one of them is in the bait set B (both strands of the bait
// sequences).  Done as a filter on the base stream: reads without a hit are blanked (every byte
// becomes 'N'), after which the normal pipeline runs on the filtered stream.  B is strand-symmetric,
// so testing the forward k-mer of each window is enough.
// ---------------------------------------------------------------------------------------
struct NewlineOp {       // positions of the read terminators
    const u8* stream; u64* out;
    struct R { u64 idx; };
    __device__ bool load(u64 idx, R& r) const { r.idx = idx; return stream[idx] == '\n'; }
    __device__ void store(u64 pos, const R& r) const { out[pos] = r.idx; }
};

constexpr int CAP_TILE = SEL_BLOCK * SEL_ITEMS;
__global__ __launch_bounds__(SEL_BLOCK) void bait_hit_kernel(const u8* __restrict__ stream, u64 n_bytes, int K,
                                                             const u64* __restrict__ baits, u64 n_baits, u8* __restrict__ hit,
                                                             u32 tiles) {
    __shared__ TileImage<CAP_TILE> img;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const u64 t0 = (u64)tile * CAP_TILE;
        stage_tile<SEL_BLOCK, CAP_TILE>(stream, n_bytes, t0, img);
#pragma unroll
        for (int i = 0; i < SEL_ITEMS; i++) {
            const int p = (int)threadIdx.x + i * SEL_BLOCK;
            u64 x;
            u8 h = 0;
            if (window_at(img, p, K, x)) {
                u64 lo = 0, hi = n_baits;
                while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (baits[mid] < x) lo = mid + 1; else hi = mid; }
                h = (lo < n_baits && baits[lo] == x) ? 1 : 0;
            }
            if (t0 + p < n_bytes) hit[t0 + p] = h;
        }
        __syncthreads();
    }
}

// one wave per read: OR of the hits over the read; blank the read in `out` when there is none
__global__ void capture_apply_kernel(const u8* __restrict__ stream, const u8* __restrict__ hit, const u64* __restrict__ ends,
                                     u64 n_reads, u8* __restrict__ out, u64* n_kept) {
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    u64 kept = 0;
    for (u64 r = wave; r < n_reads; r += nw) {
        const u64 b = r ? ends[r - 1] + 1 : 0, e = ends[r];
        bool any = false;
        for (u64 i = b + lane; i < e; i += 64) any |= hit[i] != 0;
        const bool keep = __ballot(any) != 0;
        for (u64 i = b + lane; i < e; i += 64) out[i] = keep ? stream[i] : (u8)'N';
        if (lane == 0) { out[e] = '\n'; kept += keep ? 1 : 0; }
    }
    if (lane == 0 && kept) atomicAdd(n_kept, kept);
}

int capture_filter(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, const u64* baits, uint64_t n_baits, u8* out,
                   uint64_t* n_reads, uint64_t* n_kept) {
    *n_reads = 0; *n_kept = 0;
    if (n_bytes == 0) return ZK_OK;
    if ((uintptr_t)stream & 15) return fail(c, ZK_EINVAL, "base stream must be 16-byte aligned");
    u8* hit; u64* ends;
    ZK_TRY(arena_require(c, 9 * n_bytes + (1 << 20), 9 * n_bytes + (1 << 20)));
    ZK_TRY(arena_alloc(c, n_bytes, (void**)&hit));
    ZK_TRY(arena_alloc(c, 8 * n_bytes, (void**)&ends));
    const u32 tiles = (u32)div_up(n_bytes, CAP_TILE);
    const u32 grid = tiles < (u32)c->num_cus * 8 ? tiles : (u32)c->num_cus * 8;
    hipLaunchKernelGGL(bait_hit_kernel, dim3(grid), dim3(SEL_BLOCK), 0, c->stream, stream, (u64)n_bytes, K, baits, (u64)n_baits, hit, tiles);
    ZK_HIP(c, hipGetLastError());
    NewlineOp op{stream, ends};
    uint64_t nr = 0;
    ZK_TRY(run_select(c, op, n_bytes, n_bytes, &nr));
    *n_reads = nr;
    // bytes after the last terminator (a stream that does not end in '\n') belong to no read: copy them
    ZK_HIP(c, hipMemcpyAsync(out, stream, n_bytes, hipMemcpyDeviceToDevice, c->stream));
    u64* d_kept = c->d_scalars + 11;
    ZK_HIP(c, hipMemsetAsync(d_kept, 0, sizeof(u64), c->stream));
    if (nr) {
        u64 g = div_up(nr, 4);
        if (g > (u64)c->num_cus * 16) g = (u64)c->num_cus * 16;
        hipLaunchKernelGGL(capture_apply_kernel, dim3((u32)g), dim3(256), 0, c->stream, stream, hit, ends, (u64)nr, out, d_kept);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 11, d_kept, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_kept = c->h_scalars[11];
    return check_device_error(c);
}

// ---------------------------------------------------------------------------------------
// K7: histogram of counts.  Values below HIST_DENSE go to LDS bins; the (rare) larger ones are
// appended to a list that the host folds in.
// ---------------------------------------------------------------------------------------
constexpr int HIST_DENSE = 4096;

template <typename CT>
__global__ __launch_bounds__(256) void count_hist_kernel(const CT* __restrict__ counts, u64 n, u64* __restrict__ dense,
                                                         u64* __restrict__ big, u64 big_cap, u64* big_n) {
    __shared__ u32 bins[HIST_DENSE];
    for (int i = threadIdx.x; i < HIST_DENSE; i += 256) bins[i] = 0;
    __syncthreads();
    // Counts are few distinct values (most k-mers of a sequencing run occur once, the rest near the coverage), and 64
    // LDS atomics on one word take 64 turns: peel the two commonest values of the wave, their first lane adds the lot.
    // The commonest value of all -- 1: the k-mers that a read error made -- never reaches LDS: a register per thread, one add per
    // wave at the end.  Four rows of loads are in flight at a time.
    constexpr int ROWS = 4;
    u32 ones = 0;
    const u64 step = (u64)gridDim.x * 256 * ROWS;
    for (u64 i0 = (u64)blockIdx.x * 256 * ROWS; i0 < n; i0 += step) {      // uniform trip count: every lane reaches the ballots
      u64 vv[ROWS];
#pragma unroll
      for (int q = 0; q < ROWS; q++) {
          const u64 i = i0 + (u64)q * 256 + threadIdx.x;
          vv[q] = i < n ? (u64)counts[i] : 1ull << 63;          // (past the end: a value nobody counts)
      }
#pragma unroll
      for (int q = 0; q < ROWS; q++) {
        const u64 v = vv[q];
        ones += v == 1 ? 1u : 0u;
        bool act = v != 1 && v != (1ull << 63);
        if (!__any((int)act)) continue;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const u64 m = __ballot(act && v < HIST_DENSE);
            if (m == 0) break;
            const int leader = __ffsll((long long)m) - 1;
            const u32 vl = (u32)__builtin_amdgcn_readlane((int)(u32)v, leader);
            const u64 same = __ballot(act && v == (u64)vl);
            if (__popcll(same) < 8) break;
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&bins[vl], (u32)__popcll(same));
            act = act && v != (u64)vl;
        }
        if (act) {
            if (v < HIST_DENSE) atomicAdd(&bins[v], 1u);
            else {
                const u64 slot = atomicAdd(big_n, 1ull);
                if (slot < big_cap) big[slot] = v;
            }
        }
      }
    }
    ones = wave_sum_u32(ones);
    if ((threadIdx.x & 63) == 0 && ones) atomicAdd(&bins[1], ones);
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_DENSE; i += 256)
        if (bins[i]) atomicAdd(&dense[i], (u64)bins[i]);
}

int count_hist(zk_ctx* c, const void* counts, int count_bits, uint64_t n, uint64_t* vals, uint64_t* freq,
               uint64_t cap_bins, uint64_t* n_bins) {
    *n_bins = 0;
    if (n == 0) return ZK_OK;
    const uint64_t big_cap = n < (1ull << 22) ? n : (1ull << 22);
    u64 *dense, *big;
    ZK_TRY(arena_require(c, sizeof(u64) * (HIST_DENSE + big_cap) + 4096, sizeof(u64) * (HIST_DENSE + big_cap) + 4096));
    ZK_TRY(arena_alloc(c, sizeof(u64) * HIST_DENSE, (void**)&dense));
    ZK_TRY(arena_alloc(c, sizeof(u64) * big_cap, (void**)&big));
    u64* big_n = c->d_scalars + 10;
    ZK_HIP(c, hipMemsetAsync(dense, 0, sizeof(u64) * HIST_DENSE, c->stream));
    ZK_HIP(c, hipMemsetAsync(big_n, 0, sizeof(u64), c->stream));
    u32 grid = (u32)(div_up(n, 256 * 16) < (uint64_t)c->num_cus * 8 ? div_up(n, 256 * 16) : (uint64_t)c->num_cus * 8);
    prof_begin(c, ZK_PROF_COUNT_HIST, (count_bits / 8) * n);
    if (count_bits == 32)
        hipLaunchKernelGGL((count_hist_kernel<u32>), dim3(grid), dim3(256), 0, c->stream, (const u32*)counts, (u64)n, dense, big, (u64)big_cap, big_n);
    else
        hipLaunchKernelGGL((count_hist_kernel<u64>), dim3(grid), dim3(256), 0, c->stream, (const u64*)counts, (u64)n, dense, big, (u64)big_cap, big_n);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    std::vector<u64> hd(HIST_DENSE);
    u64* h_dense = hd.data();
    ZK_HIP(c, hipMemcpyAsync(h_dense, dense, sizeof(u64) * HIST_DENSE, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 10, big_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t nb = c->h_scalars[10];
    if (nb > big_cap) return fail(c, ZK_ENOSPC, "count histogram: %llu counts >= %d exceed the side list", (unsigned long long)nb, HIST_DENSE);
    uint64_t m = 0;
    for (int v = 0; v < HIST_DENSE; v++)
        if (h_dense[v]) {
            if (m >= cap_bins) return fail(c, ZK_ENOSPC, "histogram has more than %llu bins", (unsigned long long)cap_bins);
            vals[m] = v; freq[m] = h_dense[v]; m++;
        }
    if (nb) {
        std::vector<u64> hb(nb);
        u64* h_big = hb.data();
        ZK_HIP(c, hipMemcpy(h_big, big, sizeof(u64) * nb, hipMemcpyDeviceToHost));
        std::sort(hb.begin(), hb.end());   // the few counts >= HIST_DENSE
        for (uint64_t i = 0; i < nb;) {
            uint64_t j = i;
            while (j < nb && h_big[j] == h_big[i]) j++;
            if (m >= cap_bins) return fail(c, ZK_ENOSPC, "histogram has more than %llu bins", (unsigned long long)cap_bins);
            vals[m] = h_big[i]; freq[m] = j - i; m++;
            i = j;
        }
    }
    *n_bins = m;
    return ZK_OK;
}

}  // namespace zk
