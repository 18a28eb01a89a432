// radix_sort.hip -- K3: least-significant-digit radix sort of 64-bit k-mers on gfx950.
//
// Replaces misc.radix_sort (zotmer/library/misc.py:400-424) as called from
// KmerAccumulator2.flush (zotmer/commands/kmerize.py:412-417): ascending order of the 2K
// significant bits.
//
// Shape ("onesweep"): ONE histogram kernel counts every pass's digits up front; then each pass
// reads every key once and writes it once.  Inside a pass a workgroup owns a tile of
// BLOCK*ITEMS keys held in registers in wave-striped order, ranks them with wave64 ballots
// (match-any over the digit bits + mbcnt), learns where its keys go from the tiles before it
// through a decoupled look-back (common.hpp), regroups the tile by digit in LDS and writes
// digit-contiguous runs, so stores are coalesced per run.  Tiles are numbered by a ticket, not
// by blockIdx, so the look-back never waits for a workgroup that has not started.
//
// The first pass can take its keys from a BASE STREAM instead of an array (SRC_STREAM): the
// tile is encoded in LDS (encode_tile.hpp) and the keys never exist in HBM unsorted.  That
// removes the 8 B/key write + 8 B/key read an unfused encode would cost, and lets invalid
// windows vanish without a compaction pass (dead items are simply not ranked).
//
// Algorithmic bytes per key per pass: 8 read + 8 written (+4/+4 with a 32-bit payload).
//
// The geometry (threads per workgroup, keys per thread, digit bits) is a compile-time Cfg; a few
// are instantiated and zk_tune(ZK_TUNE_SORT_VARIANT) picks one (default chosen from
// measurements on MI355X).
#include <stdlib.h>
#include "internal.hpp"
#include "encode_tile.hpp"
#include "dedupe.hpp"

namespace zk {

constexpr int MAX_RADIX = 1024;

enum { SRC_ARRAY = 0, SRC_STREAM = 1 };

// ROUNDS: the LDS regroup buffer holds TILE/ROUNDS keys and the tile leaves in that many rounds (less
// LDS per workgroup -> more workgroups per CU).  WPE: waves per SIMD the register allocator must allow.
// SEG: tiles per look-back segment (power of two; 0 = one serial chain per digit over all tiles).
// PIPE: array passes of keys run as the persistent two-stage pipeline (pass_pipe_kernel).
template <int BLOCK_, int ITEMS_, int RBITS_, int ROUNDS_ = 1, int WPE_ = 1, int SEG_ = 32, bool PIPE_ = false>
struct Cfg {
    static constexpr int BLOCK = BLOCK_, ITEMS = ITEMS_, RBITS = RBITS_, ROUNDS = ROUNDS_, WPE = WPE_, SEG = SEG_;
    static constexpr bool PIPE = PIPE_;
    static_assert(SEG == 0 || (BLOCK * ITEMS < 32768 && (SEG & (SEG - 1)) == 0), "16-bit tile counts");
    static constexpr int TILE = BLOCK * ITEMS, RADIX = 1 << RBITS, NW = BLOCK / 64;
    static constexpr int EXCH = TILE / ROUNDS, IPR = ITEMS / ROUNDS;   // slots / items per thread per round
    static_assert(ITEMS % ROUNDS == 0, "ROUNDS");
    static constexpr int DPT = (RADIX + BLOCK - 1) / BLOCK;   // digits per thread in the per-digit steps
    static_assert(ITEMS % 2 == 0 && 2 * ITEMS < 256, "ITEMS");
    static_assert(64 * ITEMS < 65536, "per-wave ranks are 16-bit");
    static_assert(RADIX <= MAX_RADIX && RBITS * MAX_PASSES >= 64, "RBITS");
};

static PassPlan make_plan(int key_bits, int rbits, int lo = 0) {
    PassPlan p;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    p.passes = (key_bits + rbits - 1) / rbits;
    int base = key_bits / p.passes, rem = key_bits % p.passes, s = lo;
    for (int i = 0; i < MAX_PASSES; i++) { p.shift[i] = 0; p.bits[i] = 0; }
    for (int i = 0; i < p.passes; i++) {
        p.bits[i] = base + (i < rem ? 1 : 0);
        p.shift[i] = s;
        s += p.bits[i];
    }
    return p;
}

struct SortArgs {
    // array source
    const u64* kin;
    const u32* vin;
    u64 n;
    int mirror_K;       // > 0: the key is rc(kin[i]) for this K (first pass and histogram of the mirror sort)
    // stream source
    const u8* stream;
    u64 n_bytes;
    int K;
    int mode;
    // stream source, record-aligned tiles (pipeline pass 0; 0 = tiles of TILE stream positions): every record is `rec`
    // bytes (rec - 1 bases + the separator); a tile takes `rpt` records, a thread one of the `cpr` 16-window chunks of a
    // record, which has `wpr` windows.  Verified by the histogram kernel before it is used (see sort_stream).
    u32 rec, rpt, cpr, wpr;
    u32 cpr_inv;        // ceil(2^32 / cpr): thread / cpr as a multiply (exact for thread < 2^16)
    // outputs
    u64* kout;
    u32* vout;
#ifdef ZK_PHASES
    int dbg_local;      // measurement (ZK_LOCAL_PASS=1): a tile's keys go to the tile's own place, grouped by digit -- no offsets waited for
#endif
    const u32* straddle; // pipeline, VAR 3: one bit per tile of this pass's input, set where a bucket of the pass before begins inside the tile
    int tags_out;       // pipeline, array source: the pass writes only the low 32 bits of a key, as a u32 array at kout (the last pass
                        // before the block dedupe: the bits above are the block's number, which the key's place says)
    // digit of this pass
    int shift;
    int bits;
    const u64* ghist;   // [RADIX] exclusive prefix of this pass's digit over all keys
    // look-back
    u16* part;          // [tiles][RADIX] 0x8000 | count of the digit in the tile (segmented scheme)
    u64* status;
    u32* ticket;
    u32 ticket_base;
    u32* xticket;       // pipeline: tile counters, one per XCD (32 words apart)
    u32 nx;             // ... how many of them are used (1 = one global tile order)
    u32 glog;           // ... log2 of the run of consecutive tiles an XCD takes at a time
    u32 epoch;
    u32* err;
    int prof_tag;       // 0: ZK_PROF_PASS_KEYS; else the tag this pass is timed under (zk_profile)
    u64* dbg;           // diagnostic build (-DZK_STAMPS) only: 8 time stamps per tile
    u64* dbg2;          // ... steps << 32 | spins of thread 0's look-back chain, per tile
};

#ifdef ZK_STAMPS
#define ZK_STAMP(k) do { if (a.dbg && threadIdx.x == 0) a.dbg[(u64)tile * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZK_STAMP(k) do { } while (0)
#endif

// Generate this thread's ITEMS keys of tile `tile` in wave-striped order.
//   array : key i of lane l of wave w is element  tile*TILE + w*64*ITEMS + i*64 + l
//   stream: the tile covers POS stream positions (POS = TILE, or TILE/2 when both strands are
//           emitted: items [0, ITEMS/2) are x, items [ITEMS/2, ITEMS) the matching rc(x))
// acgt (COUNT only): four 8-bit counters packed in a word, acgt[b] in byte b -- at most 2*ITEMS
// increments per call, so they cannot carry into each other.
template <class C, int SRC, bool PAIRS, bool COUNT>
__device__ __forceinline__ u32 load_tile(const SortArgs& a, u32 tile, TileImage<C::TILE>* img, u64 (&key)[C::ITEMS],
                                         u32 (&val)[C::ITEMS], u32& acgt) {
    constexpr int TILE = C::TILE, ITEMS = C::ITEMS, BLOCK = C::BLOCK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u32 live = 0;
    if (SRC == SRC_ARRAY) {
        const u64 base = (u64)tile * TILE + (u64)wave * (64 * ITEMS) + lane;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u64 idx = base + (u64)i * 64;
            const bool ok = idx < a.n;
            key[i] = ok ? a.kin[idx] : 0ull;
            if (PAIRS) val[i] = ok ? a.vin[idx] : 0u;
            live |= (ok ? 1u : 0u) << i;
        }
        if (a.mirror_K > 0) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) key[i] = revcomp(a.mirror_K, key[i]);
        }
    } else {
        const bool both = (a.mode == ZK_KEYS_BOTH);
        const int pos_per_tile = both ? TILE / 2 : TILE;
        const u64 t0 = (u64)tile * pos_per_tile;
        // stage only what this tile needs
        for (int c = threadIdx.x; c < pos_per_tile / 16 + 3; c += BLOCK) {
            u32 cc, vv;
            encode_chunk16(a.stream, a.n_bytes, t0 + 16ull * c, cc, vv);
            img->codes[c] = cc;
            img->valid[c] = vv;
        }
        __syncthreads();
        const int K = a.K;
        if (both) {
            constexpr int H = ITEMS / 2;
#pragma unroll
            for (int i = 0; i < H; i++) {
                const int p = wave * (64 * H) + i * 64 + lane;
                u64 x;
                const bool ok = window_at(*img, p, K, x);
                const u64 xb = revcomp(K, x);
                key[i] = x;
                key[i + H] = xb;
                live |= (ok ? 1u : 0u) << i;
                live |= (ok ? 1u : 0u) << (i + H);
                if (COUNT && ok) acgt += (1u << (8 * (u32)(x & 3))) + (1u << (8 * (u32)(xb & 3)));
            }
        } else if (ITEMS == 16) {
            // The first pass may take its keys in any order (nothing is ordered yet), so a thread takes 16
            // CONSECUTIVE windows: 8 LDS words per thread instead of 8 per window, no per-window bit reversal.
            u64 xs[16], xr[16];
            const u32 ok16 = windows16(*img, (int)threadIdx.x, K, xs, xr);
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u64 x = xs[i & 15], xb = xr[i & 15];
                key[i] = (a.mode == ZK_KEYS_CANONICAL) ? (x < xb ? x : xb) : x;
                if (COUNT && ((ok16 >> i) & 1u)) acgt += (1u << (8 * (u32)(x & 3))) + (1u << (8 * (u32)(xb & 3)));
            }
            live = ok16;
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const int p = wave * (64 * ITEMS) + i * 64 + lane;
                u64 x;
                const bool ok = window_at(*img, p, K, x);
                const u64 xb = revcomp(K, x);
                key[i] = (a.mode == ZK_KEYS_CANONICAL) ? (x < xb ? x : xb) : x;
                live |= (ok ? 1u : 0u) << i;
                if (COUNT && ok) acgt += (1u << (8 * (u32)(x & 3))) + (1u << (8 * (u32)(xb & 3)));
            }
        }
    }
    return live;
}

// ---------------------------------------------------------------------------------------
// up-front histogram of every pass's digit (one read of the keys, or of the stream)
// ---------------------------------------------------------------------------------------
struct HistArgs {
    SortArgs src;        // only the source fields are used
    PassPlan plan;
    u64* ghist;          // [MAX_PASSES][RADIX], zeroed by the host
    u64* acgt;           // [4] or null
    u32 tiles;
    u64* rec_info;       // PRE only: [0] = position of the stream's first newline (read), [1] += newline bytes,
                         // [2] += 16-byte chunks whose newlines are not exactly the record separators
    u64* sample;         // PRE only, or null: keys with (key >> sample_shift) == sample_value are also appended here ...
    u32* sample_n;       // ... += 1 each (appended while below sample_cap)
    u32 sample_cap;
    int sample_shift;
    u64 sample_value;
};

// position of the first '\n' in the head of the stream (~0 if there is none): the record length, if records are uniform
__global__ void first_newline_kernel(const u8* __restrict__ stream, u64 n_bytes, u64* out) {
    __shared__ u32 best;
    if (threadIdx.x == 0) best = 0xffffffffu;
    __syncthreads();
    const u64 lim = n_bytes < 65536 ? n_bytes : 65536;
    for (u64 i = threadIdx.x; i < lim; i += blockDim.x)
        if (stream[i] == '\n') atomicMin(&best, (u32)i);
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (best == 0xffffffffu) ? ~0ull : (u64)best;
}

// what a workgroup of the pipeline has in flight for its next tile
template <class C, int SRC> struct NextTile;
template <class C> struct NextTile<C, SRC_ARRAY> {
    u64 key[C::ITEMS];
    u32 val[C::ITEMS];          // (pairs only: never touched, hence never allocated, otherwise)
    u32 live = 0;
    template <bool PAIRS = false>
    __device__ __forceinline__ void issue(const SortArgs& a, u32 t, int tid, int wave, int lane) {
        const u64 base = (u64)t * C::TILE + (u64)wave * (64 * C::ITEMS) + lane;
        if ((u64)(t + 1) * C::TILE <= a.n) {
            // a whole tile (all but the last): one address, constant offsets, no per-key bound checks
            const u64* p = a.kin + base;
#pragma unroll
            for (int i = 0; i < C::ITEMS; i++) key[i] = p[i * 64];
            if constexpr (PAIRS) {
                const u32* q = a.vin + base;
#pragma unroll
                for (int i = 0; i < C::ITEMS; i++) val[i] = q[i * 64];
            }
            live = (C::ITEMS >= 32) ? ~0u : ((1u << C::ITEMS) - 1u);
            return;
        }
        live = 0;
#pragma unroll
        for (int i = 0; i < C::ITEMS; i++) {
            const u64 idx = base + (u64)i * 64;
            const bool ok = idx < a.n;
            key[i] = ok ? a.kin[idx] : 0ull;
            if constexpr (PAIRS) val[i] = ok ? a.vin[idx] : 0u;
            live |= (ok ? 1u : 0u) << i;
        }
    }
};
template <class C> struct NextTile<C, SRC_STREAM> {
    uint4 q0, q1;        // this thread's 16-byte chunk(s) of the tile's stream bytes
    __device__ __forceinline__ void issue(const SortArgs& a, u32 t, int tid, int wave, int lane) {
        const u64 t0 = a.rec ? (u64)t * a.rpt * a.rec : (u64)t * C::TILE;
        const u32 nch = a.rec ? (a.rpt * a.rec) / 16 + 3 : (u32)TileImage<C::TILE>::NCH;
        q0 = q1 = make_uint4(0, 0, 0, 0);
        if ((u32)tid < nch) q0 = load_chunk16(a.stream, a.n_bytes, t0 + 16ull * tid);
        if ((u32)tid + C::BLOCK < nch) q1 = load_chunk16(a.stream, a.n_bytes, t0 + 16ull * (tid + C::BLOCK));
    }
};

// PRE: stream source, one strand, 16 windows per thread made and counted on the fly (its own instantiation: the generic
// path needs the sixteen keys in registers and would spill under the 8-waves-per-SIMD register cap)
template <class C, int SRC, bool PRE = false>
__global__ __launch_bounds__(C::BLOCK, PRE ? 8 : 1) void hist_kernel(HistArgs h) {
    __shared__ u32 bins[MAX_PASSES * C::RADIX];
    __shared__ TileImage<C::TILE> img;
    for (int i = threadIdx.x; i < MAX_PASSES * C::RADIX; i += C::BLOCK) bins[i] = 0;
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __syncthreads();
    // stream source, one strand: the 16 bytes this thread stages for the NEXT tile are loaded while the current
    // tile is counted (two workgroups per CU do not hide a global round trip per tile on their own)
    constexpr bool PREFETCH = PRE;
    constexpr bool pre = PRE;
    NextTile<C, SRC_STREAM> nx;
    nx.q0 = nx.q1 = make_uint4(0, 0, 0, 0);
    if (pre && blockIdx.x < h.tiles) nx.issue(h.src, blockIdx.x, threadIdx.x, 0, 0);
    // Are the records uniform (every rec-th byte a newline and no other newline)?  rec comes from the position of the
    // stream's first newline; this kernel reads every byte anyway, so it checks: newline bytes are counted, and every
    // 16-byte chunk whose newlines are not exactly the expected separator is counted as bad.
    u32 rec = 0, m = 0, step_m = 0, nl_count = 0, bad_count = 0;
    if (PRE && h.rec_info) {
        const u64 first_nl = h.rec_info[0];
        if (first_nl < 0x7fffffffull) {
            rec = (u32)first_nl + 1u;
            m = (u32)(((u64)blockIdx.x * C::TILE + 16ull * threadIdx.x) % rec);
            step_m = (u32)(((u64)gridDim.x * C::TILE) % rec);
        }
    }
    for (u32 tile = blockIdx.x; tile < h.tiles; tile += gridDim.x) {
        u64 key[C::ITEMS];
        u32 val[C::ITEMS];
        u32 pk = 0;
        u32 live = 0;
        if (pre) {
            if constexpr (PREFETCH) {
                if (rec) {
                    const u32 w4[4] = {nx.q0.x, nx.q0.y, nx.q0.z, nx.q0.w};
                    const int sep = (int)(rec - 1u) - (int)m;          // where this chunk's separator is, if < 16
                    const u64 off = (u64)tile * C::TILE + 16ull * threadIdx.x;
                    const bool expect = sep >= 0 && sep < 16 && off + (u64)sep < h.src.n_bytes;
                    bool bad = false;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const u32 x = w4[q] ^ 0x0a0a0a0au;              // zero byte <=> newline
                        const u32 t = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // 0x80 in every zero byte, exact
                        nl_count += (u32)__popc(t);
                        const u32 want = (expect && (sep >> 2) == q) ? (0x80u << (8 * (sep & 3))) : 0u;
                        bad |= (t != want);
                    }
                    bad_count += bad ? 1u : 0u;
                    m += step_m;
                    if (m >= rec) m -= rec;
                }
                u32 cc, vv;
                encode_words16(nx.q0, cc, vv);
                img.codes[threadIdx.x] = cc; img.valid[threadIdx.x] = vv;
                if (threadIdx.x + C::BLOCK < TileImage<C::TILE>::NCH) {
                    encode_words16(nx.q1, cc, vv);
                    img.codes[threadIdx.x + C::BLOCK] = cc; img.valid[threadIdx.x + C::BLOCK] = vv;
                }
                __syncthreads();
                if (tile + gridDim.x < h.tiles) nx.issue(h.src, tile + gridDim.x, threadIdx.x, 0, 0);
                // keys are counted as they are made, never kept: few registers, many waves per SIMD
                Windows16<C::TILE> wg;
                wg.init(img, (int)threadIdx.x, h.src.K);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    u64 x, xb;
                    if (wg.get(i, x, xb)) {
                        pk += (1u << (8 * (u32)(x & 3))) + (1u << (8 * (u32)(xb & 3)));
                        const u64 kk = (h.src.mode == ZK_KEYS_CANONICAL) ? (x < xb ? x : xb) : x;
                        if (h.plan.passes == 2) {
                            // the usual plan (two passes over the top bits, then the block dedupe): no test per possible pass
                            atomicAdd(&bins[(u32)(kk >> h.plan.shift[0]) & ((1u << h.plan.bits[0]) - 1u)], 1u);
                            atomicAdd(&bins[C::RADIX + ((u32)(kk >> h.plan.shift[1]) & ((1u << h.plan.bits[1]) - 1u))], 1u);
                        } else {
#pragma unroll
                            for (int p = 0; p < MAX_PASSES; p++) {
                                if (p < h.plan.passes) {
                                    const u32 d = (u32)(kk >> h.plan.shift[p]) & ((1u << h.plan.bits[p]) - 1u);
                                    atomicAdd(&bins[p * C::RADIX + d], 1u);
                                }
                            }
                        }
                        if (h.sample && (kk >> h.sample_shift) == h.sample_value) {
                            const u32 at = atomicAdd(h.sample_n, 1u);
                            if (at < h.sample_cap) h.sample[at] = kk;
                        }
                    }
                }
                live = 0;         // nothing left for the generic counting loop below
            }
        } else {
            if constexpr (!PRE) live = load_tile<C, SRC, false, SRC == SRC_STREAM>(h.src, tile, &img, key, val, pk);
        }
        a0 += pk & 0xffu; a1 += (pk >> 8) & 0xffu; a2 += (pk >> 16) & 0xffu; a3 += pk >> 24;
#pragma unroll
        for (int i = 0; i < C::ITEMS; i++) {
            const bool lv = (live >> i) & 1u;
#pragma unroll
            for (int p = 0; p < MAX_PASSES; p++) {
                if (p < h.plan.passes) {
                    const u32 d = (u32)(key[i] >> h.plan.shift[p]) & ((1u << h.plan.bits[p]) - 1u);
                    bool act = lv;
                    if (SRC == SRC_ARRAY) {
                        // 64 consecutive elements of an array often share a digit (sorted or mirrored input: the
                        // low digits of rc(c) are the leading bases of c), and 64 LDS atomics on one word take 64
                        // turns.  Peel up to two such groups: their first lane adds the group's size.
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const u64 m = __ballot(act);
                            if (m == 0) break;
                            const int leader = __ffsll((long long)m) - 1;
                            const u32 dl = (u32)__builtin_amdgcn_readlane((int)d, leader);
                            const u64 same = __ballot(act && d == dl);
                            if (__popcll(same) < 8) break;
                            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&bins[p * C::RADIX + dl], (u32)__popcll(same));
                            act = act && d != dl;
                        }
                    }
                    if (act) atomicAdd(&bins[p * C::RADIX + d], 1u);
                }
            }
        }
        if (SRC == SRC_STREAM) __syncthreads();   // img is restaged by the next iteration
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h.plan.passes * C::RADIX; i += C::BLOCK) {
        u32 v = bins[i];
        if (v) atomicAdd(&h.ghist[i], (u64)v);
    }
    if (PRE && rec) {
        nl_count = wave_sum_u32(nl_count); bad_count = wave_sum_u32(bad_count);
        if ((threadIdx.x & 63) == 0) {
            if (nl_count) atomicAdd(&h.rec_info[1], (u64)nl_count);
            if (bad_count) atomicAdd(&h.rec_info[2], (u64)bad_count);
        }
    }
    if (h.acgt) {
        a0 = wave_sum_u32(a0); a1 = wave_sum_u32(a1); a2 = wave_sum_u32(a2); a3 = wave_sum_u32(a3);
        if ((threadIdx.x & 63) == 0) {
            if (a0) atomicAdd(&h.acgt[0], (u64)a0);
            if (a1) atomicAdd(&h.acgt[1], (u64)a1);
            if (a2) atomicAdd(&h.acgt[2], (u64)a2);
            if (a3) atomicAdd(&h.acgt[3], (u64)a3);
        }
    }
}

// exclusive prefix over the digits of every pass, in place; total key count to *n_out
// one workgroup of 256 threads; radix up to MAX_RADIX
__global__ void hist_scan_kernel(u64* ghist, int passes, int radix, u64* n_out) {
    __shared__ u64 wsum[4];
    const int t = threadIdx.x;
    const int per = radix / 256 > 0 ? radix / 256 : 1;   // radix is a power of two
    for (int p = 0; p < passes; p++) {
        u64 v[MAX_RADIX / 256];
        u64 s = 0;
#pragma unroll
        for (int j = 0; j < MAX_RADIX / 256; j++) {
            const int d = t * per + j;
            v[j] = (j < per && d < radix) ? ghist[p * radix + d] : 0;
            s += v[j];
        }
        u64 inc = wave_incl_scan_u64(s);
        if ((t & 63) == 63) wsum[t >> 6] = inc;
        __syncthreads();
        u64 off = 0;
        for (int w = 0; w < (t >> 6); w++) off += wsum[w];
        u64 run = off + inc - s;
#pragma unroll
        for (int j = 0; j < MAX_RADIX / 256; j++) {
            const int d = t * per + j;
            if (j < per && d < radix) ghist[p * radix + d] = run;
            run += v[j];
        }
        if (p == 0 && t == 255) *n_out = off + inc;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// one pass
// ---------------------------------------------------------------------------------------
template <class C>
struct PassSmem {
    union {
        u64 exch[C::EXCH];
        TileImage<C::TILE> img;
    };
    u16 cnt[C::NW][C::RADIX];
    u32 digit_off[C::RADIX];
    u64 gbase[C::RADIX];
    u32 wsum[C::NW];
    u32 ticket;
    u32 total_live;
};

// Walk back over the predecessors of `tile` for one digit: add PARTIAL counts until an INCLUSIVE
// prefix is met.  `q` points at the word of tile-1.  A word that is not published yet is polled again.
__device__ __forceinline__ u64 lookback_walk(const u64* q, u32 tile, int radix, u32 epoch, u32* err, u64* stat = nullptr) {
    u64 excl = 0;
    u32 steps = 0, polls = 0;
    for (u32 t = tile; t > 0; t--, q -= radix) {
        u64 w = ld_agent(q);
        int spins = 0;
        while (st_state(w, epoch) == 0) {
            if (++spins > ZK_SPIN_LIMIT) { atomicOr(err, ZK_DERR_SPIN_TIMEOUT | (1u << 8)); break; }
            __builtin_amdgcn_s_sleep(1);
            w = ld_agent(q);
        }
        steps++; polls += (u32)spins;
        excl += w & ZK_ST_VALUE_MASK;
        if (st_state(w, epoch) != ZK_ST_PARTIAL) break;   // INCLUSIVE (or gave up)
    }
    if (stat) *stat = ((u64)steps << 32) | polls;
    return excl;
}

// ---------------------------------------------------------------------------------------
// Segmented look-back.  Status reads are what the serial chain spends its time and its memory
// requests on (17 dependent 8-byte polls per digit per tile, measured), so:
//   * a tile publishes its digit counts as 16-bit words (bit 15 = valid; the array is cleared per pass);
//   * tiles are grouped in segments of SEG; a tile adds up the counts of the tiles before it in ITS
//     segment -- a known set, so the loads are independent and issued together;
//   * the last tile of segment g publishes the segment's counts as 64-bit epoch-tagged words (PARTIAL),
//     and the prefix through the segment (INCLUSIVE) once it knows it; every tile runs the usual
//     decoupled look-back over those SEGMENT words -- SEG times fewer of them arrive per microsecond than
//     tile words, so the walk is one to three hops.
// ---------------------------------------------------------------------------------------
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// 16-byte write-through store (one fabric write for 16 bytes; narrower sc1 stores cost one each).  A buffer
// store because that is the form whose cache policy a builtin can set (aux 16 = sc1 on gfx94x/gfx950) and
// whose completion the compiler tracks; `base` must be wave-uniform, `byte_off` is per lane.
__device__ __forceinline__ void st_agent128(void* base, u32 byte_off, u32x4 v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ u32 ld_agent32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Publish the 64 digit counts held by one wave (lane l: digit base+l) as eight 16-byte stores.
__device__ __forceinline__ void publish_counts16(u16* row, int d, u32 count, int lane) {
    const u32 v = 0x8000u | count;
    const u32 p0 = v | ((u32)__shfl_down(v, 1, 64) << 16);       // digits l, l+1
    const u32 p1 = (u32)__shfl_down(p0, 2, 64);                  // l+2, l+3
    const u32 p2 = (u32)__shfl_down(p0, 4, 64);                  // l+4, l+5
    const u32 p3 = (u32)__shfl_down(p1, 4, 64);                  // l+6, l+7
    if ((lane & 7) == 0) { u32x4 w = {p0, p1, p2, p3}; st_agent128(row, (u32)d * 2u, w); }
}
__device__ __forceinline__ u16 ld_agent16(const u16* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent16(u16* p, u16 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int SEG, int RADIX, bool PUBLISH = true>
__device__ __forceinline__ u64 lookback_segmented(const SortArgs& a, u32 tile, int d, u32 count, u64* stat) {
    const u32 seg = tile / SEG, pos = tile % SEG;
    u16* row = a.part + (u64)tile * RADIX + d;
    if (PUBLISH) st_agent16(row, (u16)(0x8000u | count));
    u64* sw = a.status + (u64)seg * RADIX + d;          // this segment's word; earlier segments lie below it
    u64 bw = (seg > 0) ? ld_agent(sw - RADIX) : st_pack(ZK_ST_INCLUSIVE, a.epoch, 0);   // first hop, issued early
    u32 sum = 0, polls = 0;
    for (u32 i0 = 1; i0 <= pos; i0 += 8) {       // pos is uniform over the workgroup
        u16 w[8];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = (i0 + k <= pos) ? ld_agent16(row - (long long)(i0 + k) * RADIX) : (u16)0x8000u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            u16 x = w[k];
            int spins = 0;
            while (!(x & 0x8000u)) {
                if (++spins > ZK_SPIN_LIMIT) { atomicOr(a.err, ZK_DERR_SPIN_TIMEOUT | (2u << 8)); break; }
                __builtin_amdgcn_s_sleep(1);
                x = ld_agent16(row - (long long)(i0 + k) * RADIX);
            }
            polls += (u32)spins;
            sum += x & 0x7fffu;
        }
    }
    const bool last = (pos == SEG - 1);
    if (last && seg > 0) st_agent(sw, st_pack(ZK_ST_PARTIAL, a.epoch, sum + count));
    // decoupled look-back over the segment words: PARTIAL = that segment's total, INCLUSIVE = prefix through it
    u64 base = 0;
    u32 hops = 0;
    const u64* q = sw - RADIX;
    for (u32 g = seg; g > 0; g--, q -= RADIX) {
        u64 w = (g == seg) ? bw : ld_agent(q);
        int spins = 0;
        while (st_state(w, a.epoch) == 0) {
            if (++spins > ZK_SPIN_LIMIT) { atomicOr(a.err, ZK_DERR_SPIN_TIMEOUT | (4u << 8)); break; }
            __builtin_amdgcn_s_sleep(1);
            w = ld_agent(q);
        }
        polls += (u32)spins; hops++;
        base += w & ZK_ST_VALUE_MASK;
        if (st_state(w, a.epoch) != ZK_ST_PARTIAL) break;
    }
    if (last) st_agent(sw, st_pack(ZK_ST_INCLUSIVE, a.epoch, base + sum + count));
    if (stat) *stat = ((u64)hops << 32) | polls;
    return base + sum;
}

// match-any over the digit bits: on return (plo, phi) = the live lanes of the wave that hold the same digit as this lane.
// All RBITS bits are always tested (bits above the pass's width are zero in every lane, so they change nothing) -- no
// data- or pass-dependent branch.  Four vector instructions per bit: the bit as 0 / ~0 (v_bfe_i32, kept opaque: left to
// itself the compiler rebuilds it from a shift, a sign compare and an arithmetic shift), its ballot, and per half ONE
// three-input boolean op  peers &= ~(ballot ^ bit)  (v_bitop3_b32, truth table 0x90 for a & ~(b ^ c)).
template <int RBITS>
__device__ __forceinline__ void match_digit(u32 d, u64 live_mask, u32& plo, u32& phi) {
    plo = (u32)live_mask; phi = (u32)(live_mask >> 32);
#pragma unroll
    for (int b = 0; b < RBITS; b++) {
        int B;
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(B) : "v"(d), "n"(b));
        const u64 m = __ballot(B != 0);
        plo = __builtin_amdgcn_bitop3_b32(plo, (u32)m, (u32)B, 0x90);
        phi = __builtin_amdgcn_bitop3_b32(phi, (u32)(m >> 32), (u32)B, 0x90);
    }
}

template <class C, int SRC, bool PAIRS>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void pass_kernel(SortArgs a) {
    constexpr int BLOCK = C::BLOCK, ITEMS = C::ITEMS, RADIX = C::RADIX, NW = C::NW, DPT = C::DPT;
    __shared__ PassSmem<C> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const u32 tile = take_ticket(a.ticket, &sm.ticket) - a.ticket_base;
    ZK_STAMP(0);

    u64 key[ITEMS];
    u32 val[ITEMS];
    u32 unused = 0;
    const u32 live = load_tile<C, SRC, PAIRS, false>(a, tile, &sm.img, key, val, unused);
    const u32 dmask = (1u << a.bits) - 1u;

    u16* mycnt = sm.cnt[wave];
    for (int d = lane; d < RADIX; d += 64) mycnt[d] = 0;
    __syncthreads();
#ifdef ZK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZK_STAMP(1);     // keys have arrived
#endif

    // thread t owns digits t*DPT .. t*DPT+DPT-1 in the per-digit steps
    u32 tcount[DPT];
    u32 dig_excl = 0;

    // ---- rank inside the wave ------------------------------------------------------------
    u32 rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const bool lv = (live >> i) & 1u;
        const u32 d = (u32)(key[i] >> a.shift) & dmask;
        // match-any over the digit bits: peers = live lanes of the wave holding the same digit.
        // All RBITS bits are always tested (bits above a.bits are zero in every lane, so they
        // change nothing) -- no data-dependent or pass-dependent branch in the loop.
        u32 plo, phi;
        match_digit<C::RBITS>(d, __ballot(lv), plo, phi);
        const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const u32 npeer = (u32)__popc(plo) + (u32)__popc(phi);
        const u32 pre = lv ? (u32)mycnt[d] : 0u;
        rank[i] = pre + below;
        if (lv && below == npeer - 1) mycnt[d] = (u16)(pre + npeer);   // highest peer lane updates
    }
    ZK_STAMP(2);         // this wave ranked
    __syncthreads();
    ZK_STAMP(3);         // every wave ranked

    // ---- per digit: exclusive scan over the waves, tile total ---------------------------------
    u32 tsum2 = 0;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
        const int d = tid * DPT + j;
        u32 acc = 0;
        if (d < RADIX) {
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const u32 t = sm.cnt[w][d];
                sm.cnt[w][d] = (u16)acc;
                acc += t;
            }
        }
        tcount[j] = acc;
        tsum2 += acc;
    }
    // exclusive scan over the digits
    const u32 inc = wave_incl_scan_u32(tsum2);
    if (lane == 63) sm.wsum[wave] = inc;
    __syncthreads();
    u32 woff = 0;
    for (int w = 0; w < wave; w++) woff += sm.wsum[w];
    dig_excl = woff + inc - tsum2;
    if (tid == BLOCK - 1) sm.total_live = woff + inc;

    // ---- decoupled look-back, one chain per digit ----------------------------------------------
    ZK_STAMP(4);
#pragma unroll
    for (int j = 0; j < DPT; j++) {
        const int d = tid * DPT + j;
        if (d < RADIX) {
            u64 excl = 0;
#ifdef ZK_STAMPS
            u64* stat = (a.dbg2 && tid == 0) ? a.dbg2 + tile : nullptr;
#else
            u64* stat = nullptr;
#endif
            if (C::SEG > 0) {
                excl = lookback_segmented<(C::SEG > 0 ? C::SEG : 1), RADIX>(a, tile, d, tcount[j], stat);
            } else {
                u64* st = a.status + (u64)tile * RADIX + d;
                if (tile == 0) {
                    st_agent(st, st_pack(ZK_ST_INCLUSIVE, a.epoch, tcount[j]));
                } else {
                    st_agent(st, st_pack(ZK_ST_PARTIAL, a.epoch, tcount[j]));
                    excl = lookback_walk(st - RADIX, tile, RADIX, a.epoch, a.err, stat);
                    st_agent(st, st_pack(ZK_ST_INCLUSIVE, a.epoch, excl + tcount[j]));
                }
            }
            sm.digit_off[d] = dig_excl;
            sm.gbase[d] = a.ghist[d] + excl - dig_excl;
            dig_excl += tcount[j];
        }
    }
    ZK_STAMP(5);         // this thread's look-back chain done
    __syncthreads();
    ZK_STAMP(6);         // every chain done

    // ---- regroup the tile by digit in LDS, EXCH slots at a time ------------------------------
    constexpr int EXCH = C::EXCH, IPR = C::IPR;
    u32 lpos[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 d = (u32)(key[i] >> a.shift) & dmask;
        lpos[i] = ((live >> i) & 1u) ? sm.digit_off[d] + sm.cnt[wave][d] + rank[i] : ~0u;
    }
    const u32 total = sm.total_live;   // written before the barrier that ended the look-back
    u32* exv = reinterpret_cast<u32*>(sm.exch);
#pragma unroll
    for (int r = 0; r < C::ROUNDS; r++) {
        const u32 lo = (u32)r * EXCH;
        if (r > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; i++)
            if (lpos[i] - lo < (u32)EXCH) sm.exch[lpos[i] - lo] = key[i];
        __syncthreads();
        u64 gpos[IPR];
        // four slots at a time: their LDS reads are independent (slots past the live count hold stale but readable
        // keys); only the global store is predicated
        constexpr int G = (IPR % 4 == 0) ? 4 : 1;
#pragma unroll
        for (int i0 = 0; i0 < IPR; i0 += G) {
            u64 kk[G];
#pragma unroll
            for (int g = 0; g < G; g++) kk[g] = sm.exch[tid + (i0 + g) * BLOCK];
#pragma unroll
            for (int g = 0; g < G; g++) gpos[i0 + g] = sm.gbase[(u32)(kk[g] >> a.shift) & dmask] + lo + (tid + (i0 + g) * BLOCK);
#pragma unroll
            for (int g = 0; g < G; g++)
                if (lo + tid + (i0 + g) * BLOCK < total && gpos[i0 + g] < a.n) a.kout[gpos[i0 + g]] = kk[g];          // (< n: a look-back that gave up -- ZK_DERR_SPIN_TIMEOUT -- left offsets that mean nothing)
        }
        if (PAIRS) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ITEMS; i++)
                if (lpos[i] - lo < (u32)EXCH) exv[lpos[i] - lo] = val[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < IPR; i++) {
                const u32 s = tid + i * BLOCK;
                if (lo + s < total && gpos[i] < a.n) a.vout[gpos[i]] = exv[s];
            }
        }
    }
#ifdef ZK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ZK_STAMP(7);         // keys stored
#endif
}

// ---------------------------------------------------------------------------------------
// Collapse pass (zk_kmerize, canonical keys; see pipeline.hip::kmerize_full).
//
// Input: the keys ordered by their low `shift` bits, where 2^shift is within a factor of a tile of the number of keys -- so
// all copies of a k-mer (they share every bit) sit within a few dozen slots of each other, i.e. nearly always in ONE tile,
// interleaved with the few other k-mers that share their low bits.  The tile is ranked by the NEXT digit exactly as a sort
// pass would and parked in LDS grouped by that digit: inside a digit group the keys are still in input order, hence ordered
// by shift + bits low bits, hence equal keys are neighbours.  Instead of scattering the 8192 keys to the digit's global
// place, the runs are counted right there and the tile writes one word (key << pack | run length) per run, tiles one after
// the other (a one-word look-back per tile; no global histogram of this digit is needed at all).
//
// The output is NOT ordered by the digit across tiles -- it is tile-major -- so the upper-bit passes that follow start at
// bit `shift`, not shift + bits: inside (tile, digit group) the words ascend by their low `shift` bits and the tiles
// partition the range of those bits in ascending order, so for any digit value the stable pass collects tile 0's group,
// tile 1's group, ... = ascending low bits.  This needs the FIRST of those passes to use exactly this digit (a narrower one
// would concatenate two groups of one tile, whose low bits overlap): the caller takes `bits` from sort_first_bits.
// LSD invariant kept; a run cut by a tile edge (or a k-mer whose copies were not neighbours) yields two words with the
// same key, which reduce_by_key adds up after the sort, as before.
// `split`: runs are also cut every 512 slots so that a length always fits `pack` < 14 bits.
//
// Measured alternatives (config 2, 6.2 G keys; this version 21 ms): a persistent variant that prefetches the next tile's keys
// -- with the one-word look-back all resident workgroups fall into lock step (2.5x slower); with per-workgroup output
// regions and a compacting copy instead of the look-back no faster (the kernel is bound by its ~1800 vector instructions
// per thread, not by the loads); counting in an LDS hash table (compare-and-swap claims, adds count; output in input
// order, no ranking at all) 30-35 ms: the copies of a k-mer sit in the same 64 lanes, so every atomic instruction carries
// several same-address conflicts (SQ_LDS_BANK_CONFLICT 43 % of the kernel's cycles).
// ---------------------------------------------------------------------------------------
template <int RBITS>
struct CollapseSmem {
    static constexpr int BLOCK = 512, ITEMS = 16, TILE = BLOCK * ITEMS, RADIX = 1 << RBITS, NW = BLOCK / 64, CHUNKS = TILE / 64;
    u64 exch[TILE];
    u16 cnt[NW][RADIX];
    u32 digit_off[RADIX];
    u64 mask[CHUNKS];           // head flags of slots [64 q, 64 q + 64)
    u32 hbase[CHUNKS];          // heads before chunk q
    u32 nexth[CHUNKS];          // first head at or after slot 64 (q + 1)
    u32 wsum[NW];
    u32 ticket;
    u32 total_live;
    u32 heads;
    u64 gbase;
};

struct CollapseArgs {
    const u64* kin;
    u64 n;
    u64* out;
    u64 cap;
    int shift, bits, pack, split;
    u64* status;
    u32* ticket;
    u32 ticket_base;
    u32 epoch;
    u32* err;
    u64* d_total;
    u32 tiles;
};

template <int RBITS>
__global__ __launch_bounds__(512, 4) void collapse_kernel(CollapseArgs a) {
    using S = CollapseSmem<RBITS>;
    constexpr int BLOCK = S::BLOCK, ITEMS = S::ITEMS, TILE = S::TILE, RADIX = S::RADIX, NW = S::NW, CHUNKS = S::CHUNKS;
    static_assert(RADIX <= BLOCK, "one digit per thread");
    __shared__ S sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 tile = take_ticket(a.ticket, &sm.ticket) - a.ticket_base;
    // as load_tile: a wave takes 64 * ITEMS consecutive keys (the ranks below number them wave by wave, row by row)
    const u64 base = (u64)tile * TILE + (u64)wave * (64 * ITEMS) + lane;
    const u32 dmask = (1u << a.bits) - 1u;

    u64 key[ITEMS];
    u32 live = 0;
    if ((u64)(tile + 1) * TILE <= a.n) {
        const u64* p = a.kin + base;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) key[i] = p[i * 64];
        live = (1u << ITEMS) - 1u;
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u64 g = base + (u64)i * 64;
            key[i] = 0;
            if (g < a.n) { key[i] = a.kin[g]; live |= 1u << i; }
        }
    }
    u16* mycnt = sm.cnt[wave];
    for (int q = lane; q < RADIX / 8; q += 64) reinterpret_cast<uint4*>(mycnt)[q] = make_uint4(0, 0, 0, 0);

    // ---- rank inside the wave (see pass_kernel) -------------------------------------------------
    u32 rank2[ITEMS / 2];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const bool lv = (live >> i) & 1u;
        const u32 d = (u32)(key[i] >> a.shift) & dmask;
        u32 plo, phi;
        match_digit<RBITS>(d, __ballot(lv), plo, phi);
        const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
        const u32 npeer = (u32)__popc(plo) + (u32)__popc(phi);
        const u32 pre = lv ? (u32)mycnt[d] : 0u;
        if (i & 1) rank2[i / 2] |= (pre + below) << 16; else rank2[i / 2] = pre + below;
        if (lv && below == npeer - 1) mycnt[d] = (u16)(pre + npeer);
    }
    __syncthreads();
    // ---- per digit: exclusive scan over the waves, then over the digits ----------------------------
    u32 acc = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const u32 t = sm.cnt[w][tid];
            sm.cnt[w][tid] = (u16)acc;
            acc += t;
        }
    }
    const u32 inc = wave_incl_scan_u32(acc);
    if (lane == 63) sm.wsum[wave] = inc;
    __syncthreads();
    {
        u32 woff = 0;
        for (int w = 0; w < wave; w++) woff += sm.wsum[w];
        if (tid < RADIX) sm.digit_off[tid] = woff + inc - acc;
        if (tid == BLOCK - 1) sm.total_live = woff + inc;
    }
    __syncthreads();
    // ---- park, grouped by digit ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 d = (u32)(key[i] >> a.shift) & dmask;
        if ((live >> i) & 1u) sm.exch[sm.digit_off[d] + sm.cnt[wave][d] + ((rank2[i / 2] >> (16 * (i & 1))) & 0xffffu)] = key[i];
    }
    __syncthreads();
    const u32 total = sm.total_live;
    // ---- heads of the runs: slot s = 512 i + tid, chunk q = 8 i + wave --------------------------------
    u32 headbits = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 s = (u32)i * BLOCK + tid;
        const u64 k = sm.exch[s];
        const u64 prev = sm.exch[s ? s - 1 : 0];
        key[i] = k;
        const bool head = s < total && (s == 0 || k != prev || (a.split && tid == 0));
        const u64 m = __ballot(head);
        if (lane == 0) sm.mask[i * NW + wave] = m;
        headbits |= (head ? 1u : 0u) << i;
    }
    __syncthreads();
    // ---- heads before every chunk, first head after it; the tile's place in the output -----------------------
    if (wave == 0) {
        static_assert(CHUNKS == 128, "two chunks per lane");
        const u64 m0 = sm.mask[2 * lane], m1 = sm.mask[2 * lane + 1];
        const u32 p0 = (u32)__popcll(m0), p1 = (u32)__popcll(m1);
        const u32 in2 = wave_incl_scan_u32(p0 + p1);
        sm.hbase[2 * lane] = in2 - p0 - p1;
        sm.hbase[2 * lane + 1] = in2 - p1;
        // first head in the chunks AFTER q (suffix minimum over the chunks' first heads; `total` if there is none)
        const u32 f0 = m0 ? (u32)(2 * lane) * 64 + (u32)__builtin_ctzll(m0) : 0xffffffffu;
        const u32 f1 = m1 ? (u32)(2 * lane + 1) * 64 + (u32)__builtin_ctzll(m1) : 0xffffffffu;
        u32 suf = f0 < f1 ? f0 : f1;          // inclusive suffix minimum over the lanes' pairs
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = (u32)__shfl_down((int)suf, o, 64);
            if (lane + o < 64 && t < suf) suf = t;
        }
        u32 after = (u32)__shfl_down((int)suf, 1, 64);          // over the pairs after this lane's
        if (lane == 63) after = 0xffffffffu;
        const u32 a1 = after < total ? after : total;
        sm.nexth[2 * lane + 1] = a1;
        sm.nexth[2 * lane] = f1 < a1 ? f1 : a1;
        const u32 m = (u32)__builtin_amdgcn_readlane((int)in2, 63);
        const u64 excl = lookback_exclusive(a.status, tile, (u64)m, a.epoch, a.err);
        if (lane == 0) {
            sm.heads = m;
            sm.gbase = excl;
            if (tile == a.tiles - 1) *a.d_total = excl + m;
            if (excl + m > a.cap) atomicOr(a.err, ZK_DERR_CAPACITY);
        }
    }
    __syncthreads();
    const u64 gbase = sm.gbase;
    if (gbase + sm.heads > a.cap) return;
    // ---- one word per run ---------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32 s = (u32)i * BLOCK + tid;
        const u32 q = (u32)i * NW + wave;
        const u64 m = sm.mask[q];
        const u64 rest = (lane < 63) ? (m >> (lane + 1)) : 0ull;
        const u32 nx = rest ? s + 1 + (u32)__builtin_ctzll(rest) : sm.nexth[q];
        const u32 j = sm.hbase[q] + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        if ((headbits >> i) & 1u) a.out[gbase + j] = (key[i] << a.pack) | (u64)(nx - s);
    }
}

// ---------------------------------------------------------------------------------------
// Block dedupe (zk_kmerize, canonical keys; see pipeline.hip::kmerize_full): counting AND finishing the sort in LDS.
//
// After LSD passes over the TOP b bits of the keys, the keys that share those bits form a block -- and all copies of a k-mer
// lie in one block (they share every bit).  While a block is small (n / 2^b keys; 23.7 K after two passes on config 2) one
// workgroup counts it in an LDS hash table: a compare-and-swap claims an entry for a key's remaining bits (its tag), an add
// counts the copy.  The block's entries (~3 K distinct tags) are then sorted right there: a counting sort on the tag's top byte
// (LDS counters), and inside each byte's group of a dozen entries the place is the number of smaller tags.  The block's
// distinct k-mers therefore leave the kernel SORTED, and the blocks are in the order of their top bits: the counted list
// needs no further sort pass at all.
// The copies of a k-mer are spread over the whole block (unlike in the tile-local table variant that was measured for
// collapse_kernel, where they sit in the same 64 lanes), so the atomics rarely collide.
// Words (key << pack | count) go to the block's own place in `out` (its input offset: never more words than keys);
// dedupe_unpack_kernel moves them together and splits them into keys and counts.  A count beyond the field leaves the field 0
// and goes to a side list that patches the count afterwards.  A table that fills up (more than ~6 K distinct keys in a block:
// little duplication) raises a flag and the caller sorts the keys the long way -- the result never depends on the table.
// ---------------------------------------------------------------------------------------
// TAG32: a tag fits 32 bits: entries of 4 + 4 bytes.  The blocks are not of one size -- a canonical k-mer more often starts with
// A than with T (it is the smaller strand), so the sizes spread from ~0 to 2 x the mean with the first bases; the table is sized
// for the big ones: one 1024-thread workgroup per CU.
template <bool TAG32>
struct DedupeSmem {
    static constexpr int BLOCK = 1024, ITEMS = 8, TILE = BLOCK * ITEMS, NW = BLOCK / 64, ALL = TAG32 ? 12288 : 6144, SPT = ALL / BLOCK, NB = 256;
    // a wave's side list: what one tile can add at worst (64 * ITEMS) on top of what is left standing after a tile (SIDE_KEEP)
    static constexpr int SIDE_KEEP = 128, SIDE = SIDE_KEEP + 64 * ITEMS;
    typedef typename std::conditional<TAG32, u32, u64>::type E;
    E keys[ALL];             // tags (after the count: the entries again, grouped by their top byte)
    u32 cnt[ALL];
    E side[NW][SIDE];
    u32 bc[NB];              // entries per top byte of the tag
    u32 bbase[NB + 1];       // ... before it
    u32 bfill[NB];
    u32 ticket;
};

// cuts[v] = first index whose key >> tag_bits is >= v, v = 0 .. blocks
__global__ void dedupe_cuts_kernel(const u64* __restrict__ k, u64 n, int tag_bits, u32 blocks, u64* __restrict__ cuts) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > blocks) return;
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((k[mid] >> tag_bits) < (u64)v) lo = mid + 1; else hi = mid;
    }
    cuts[v] = lo;
}

// ticket -> block: the blocks in order, or (second chance of the blocks dedupe2_kernel declined) the ones on a list
__device__ __forceinline__ u32 dedupe_block_of(const DedupeArgs& a, u32 ticket) { return a.list ? a.list[ticket] : ticket; }

// what a workgroup carries from one block to the next: the block it is about to count (its ticket), with the first tile of its keys
// already asked for -- the ticket, the bounds and those keys travel while the previous block is being sorted and written
template <int ITEMS>
struct DedupeNext {
    u32 chunk;
    u64 lo, hi;
    u64 key[ITEMS];
};

template <bool TAG32, bool TAGIN>
__device__ __forceinline__ void dedupe_block(const DedupeArgs& a, DedupeSmem<TAG32>& sm, DedupeNext<DedupeSmem<TAG32>::ITEMS>& st, u32 (&ph)[8], u32& tlast) {
    using S = DedupeSmem<TAG32>;
    using E = typename S::E;
    constexpr int BLOCK = S::BLOCK, ITEMS = S::ITEMS, TILE = S::TILE, ALL = S::ALL, SPT = S::SPT, NB = S::NB;
    constexpr E EMPTY = (E)~(E)0;            // no entry.  A 64-bit tag never has all its bits set; a 32-bit one may: see `home`
    constexpr u32 HS = TAG32 ? ALL - 1 : ALL;          // ... then the last entry belongs to the all-ones tag alone
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 chunk = dedupe_block_of(a, st.chunk);
    const u64 lo = st.lo, hi = st.hi;
    const u32 maxc = (1u << a.pack) - 1u;
    // whole tiles: one address, constant offsets; the cut last tile: per-key bounds
    auto load = [&](u64 base, u64 end, u64 (&k)[ITEMS]) {
        if constexpr (TAGIN) {
            // (a tag is a whole key as far as the table goes: the block's bits are added when the words are written)
            if (base + TILE <= end) {
                // four tags per load (16 bytes a lane, a kilobyte a wave instruction; which thread takes which key is the table's
                // business alone); a block starts wherever it starts: the loads are 4-byte aligned, no more
                struct __attribute__((packed, aligned(4))) Tag4 { u32 a, b, c, d; };
                static_assert(ITEMS % 4 == 0, "whole quads");
#pragma unroll
                for (int i = 0; i < ITEMS / 4; i++) {
                    const Tag4 q = *reinterpret_cast<const Tag4*>(a.tin + base + (u64)i * (4 * BLOCK) + 4 * tid);
                    k[4 * i] = q.a; k[4 * i + 1] = q.b; k[4 * i + 2] = q.c; k[4 * i + 3] = q.d;
                }
            } else {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) {
                    const u64 g = base + (u64)i * BLOCK + tid;
                    k[i] = g < end ? (u64)a.tin[g] : ~0ull;
                }
            }
        } else if (base + TILE <= end) {
            const u64* p = a.kin + base + tid;
#pragma unroll
            for (int i = 0; i < ITEMS; i++) k[i] = p[i * BLOCK];
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const u64 g = base + (u64)i * BLOCK + tid;
                k[i] = g < end ? a.kin[g] : ~0ull;
            }
        }
    };
    if (tid == 0) sm.ticket = atomicAdd(a.counter, 1u);          // the block after this one: read after the next barrier
    if (hi <= lo) {
        if (tid == 0) a.nwords[chunk] = 0;
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        __syncthreads();
        st.chunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
        st.lo = st.hi = 0;
        if (st.chunk < a.chunks) { const u32 nb = dedupe_block_of(a, st.chunk); st.lo = a.cuts[nb]; st.hi = a.cuts[nb + 1]; }
        if (st.hi > st.lo) load(st.lo, st.hi, st.key);
        return;
    }
    for (int q = tid; q < ALL * (int)sizeof(E) / 16; q += BLOCK) reinterpret_cast<uint4*>(sm.keys)[q] = make_uint4(~0u, ~0u, ~0u, ~0u);
    for (int q = tid; q < ALL / 4; q += BLOCK) reinterpret_cast<uint4*>(sm.cnt)[q] = make_uint4(0, 0, 0, 0);
    if (tid < NB) { sm.bc[tid] = 0; sm.bfill[tid] = 0; }
    __syncthreads();
    DD_PHASE(0);          // table cleared
    const u32 nchunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    u64 nlo = 0, nhi = 0;
    if (nchunk < a.chunks) { const u32 nb = dedupe_block_of(a, nchunk); nlo = a.cuts[nb]; nhi = a.cuts[nb + 1]; }
    u32 bad = 0;
    // The kernel is bound by its instruction count (188 per key with several keys probing at once, 88 with one tight probing
    // loop per key -- a loop runs as long as the unluckiest of its 64 lanes).  So the common case has NO loop and no branch:
    // one compare-and-swap at the key's home entry, the count added as 1 or 0 (adding 0 to another key's entry harms nobody);
    // a key that finds another key at home goes to the wave's side list (its place from a ballot, no atomic), and the lists --
    // about a tenth of the distinct keys with all their copies -- are inserted by linear probing afterwards, full wavefronts.
    // (An order-preserving "hash" -- the tag scaled to the table -- would leave the table sorted, but the error variants of a
    // k-mer differ from it in a few low bits and all want the same entry: 45 ms instead of 17.)
    const u64 tmask = (1ull << a.tag_bits) - 1;
    u32 nside = 0;          // entries in this wave's side list (the same in every lane)
    auto home = [&](E e) -> u32 {
        // the all-ones 32-bit tag (= the empty marker) has the last entry to itself: there the swap of "empty" for "empty"
        // succeeds and leaves the word as it is; no other key is ever sent there
        u32 x;
        if constexpr (TAG32) x = (u32)e * 0x9E3779B1u; else x = ((u32)((u64)e >> 24) ^ ((u32)e * 0x85EBCA6Bu)) * 0x9E3779B1u;
        if (TAG32 && e == EMPTY) return HS;
        return (u32)(((u64)x * HS) >> 32);
    };
    auto cas = [&](u32 h, E e) -> E {
        if constexpr (TAG32) return atomicCAS(&sm.keys[h], EMPTY, e);
        else return (E)atomicCAS(reinterpret_cast<unsigned long long*>(&sm.keys[h]), (unsigned long long)EMPTY, (unsigned long long)e);
    };
    auto drain = [&]() {          // the wave's side list into the table by linear probing, 64 entries at a time
        for (u32 i = (u32)lane; i < nside; i += 64) {
            const E e = sm.side[wave][i];
            u32 h = home(e) + 1;          // its home entry is taken: that is why it is here
            h = h == HS ? 0u : h;
            int p = 0;
            for (; p < ALL; p++) {
                const E old = cas(h, e);
                if (old == EMPTY || old == e) break;
                h = h + 1 == HS ? 0u : h + 1;
            }
            if (p < ALL) atomicAdd(&sm.cnt[h], 1u); else bad = 1;
        }
        nside = 0;
    };
    // (Measured: the eight compare-and-swaps of a tile issued back to back before any answer is used -- 24.6 ms against 21.3: the
    // insert is bound by the LDS atomic unit's throughput (two atomics per key, ~47 K per block), not by the round trips.)
    auto insert = [&](u64 k, bool valid) {
        const E e = (E)(k & tmask);
        const u32 h = home(e);
        const E old = valid ? cas(h, e) : e;          // (a plain read first, the swap only for the lanes that see "empty": no faster)
        const bool ok = old == EMPTY || old == e;
        atomicAdd(&sm.cnt[h], (ok && valid) ? 1u : 0u);
        const u64 m = __ballot(!ok);
        if (m) {
            if (!ok) sm.side[wave][nside + popc_below(m)] = e;
            nside += (u32)__popcll(m);
        }
    };
    u64 key[ITEMS], nk[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) key[i] = st.key[i];          // the first tile was asked for during the previous block
    for (u64 base = lo; base < hi; base += TILE) {
        if (base + TILE < hi) load(base + TILE, hi, nk);
        if (base + TILE <= hi) {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) insert(key[i], true);
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; i++) insert(key[i], key[i] != ~0ull);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) key[i] = nk[i];
        if (nside > (u32)S::SIDE_KEEP || base + TILE >= hi) drain();
    }
    DD_PHASE(1);          // keys inserted
    st.chunk = nchunk; st.lo = nlo; st.hi = nhi;
    if (nhi > nlo) load(nlo, nhi, st.key);          // the next block's first tile travels while this one is sorted and written
    const int any_bad = __syncthreads_or((int)bad);
    DD_PHASE(2);          // ... every wave done
    if (any_bad) {
        // the table filled up (a block with more distinct keys than it holds): the block goes on the list of those the host
        // counts by sorting; only when that list is full is the whole run given up
        if (tid == 0) {
            const u32 at = atomicAdd(a.n_bad, 1u);
            if (at < a.bad_cap) a.bad[at] = chunk; else atomicOr(a.flags, 1u);
            a.nwords[chunk] = 0;
        }
        if (a.sub && tid < 64) a.sub[(u64)chunk * 64 + tid] = 0;
        return;
    }
    // ---- the block's entries, sorted: a counting sort on the tag's top byte, then ranks inside each byte's group ---------
    // thread t takes the entries t, t + BLOCK, ... into registers; the table's memory then takes them back grouped
    E et[SPT];
    u32 ec[SPT];
    const int bsh = a.tag_bits > 8 ? a.tag_bits - 8 : 0;
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        et[j] = sm.keys[tid + j * BLOCK];
        ec[j] = sm.cnt[tid + j * BLOCK];
        if (ec[j]) atomicAdd(&sm.bc[(u32)((u64)et[j] >> bsh) & (NB - 1)], 1u);
    }
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
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        if (ec[j]) {
            const u32 b = (u32)((u64)et[j] >> bsh) & (NB - 1);
            const u32 p = sm.bbase[b] + atomicAdd(&sm.bfill[b], 1u);
            sm.keys[p] = et[j];
            sm.cnt[p] = ec[j];
        }
    }
    __syncthreads();
    DD_PHASE(4);          // grouped by top byte
    const u64 hi_part = (u64)chunk << a.tag_bits;          // the bits every key of the block has above its tag
    for (u32 i = (u32)tid; i < total; i += BLOCK) {
        const E mine = sm.keys[i];
        const u32 b = (u32)((u64)mine >> bsh) & (NB - 1);
        const u32 g0 = sm.bbase[b], g1 = sm.bbase[b + 1];
        u32 rank = 0;
        for (u32 q = g0; q < g1; q++) rank += sm.keys[q] < mine ? 1u : 0u;
        const u32 c = sm.cnt[i];
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

// Persistent: one workgroup per CU (the table takes most of its LDS) draws the blocks from a counter -- in order, not strided:
// the sizes go with the first bases, a stride of the grid would give one workgroup all the big ones.
template <bool TAG32, bool TAGIN = false>
__global__ __launch_bounds__(1024, 4) void dedupe_kernel(DedupeArgs a) {
    using S = DedupeSmem<TAG32>;
    static_assert(TAG32 || !TAGIN, "32-bit tags in, 32-bit tags in the table");
    __shared__ S sm;
    DedupeNext<S::ITEMS> st;
    if (threadIdx.x == 0) sm.ticket = atomicAdd(a.counter, 1u);
    __syncthreads();
    st.chunk = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    st.lo = st.hi = 0;
    if (st.chunk < a.chunks) { const u32 nb = dedupe_block_of(a, st.chunk); st.lo = a.cuts[nb]; st.hi = a.cuts[nb + 1]; }
#pragma unroll
    for (int i = 0; i < S::ITEMS; i++) {
        const u64 g = st.lo + (u64)i * S::BLOCK + threadIdx.x;
        if constexpr (TAGIN) st.key[i] = g < st.hi ? (u64)a.tin[g] : ~0ull;
        else st.key[i] = g < st.hi ? a.kin[g] : ~0ull;
    }
    __syncthreads();          // the ticket word is free again
    u32 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 tlast = a.dbg ? (u32)__builtin_amdgcn_s_memtime() : 0u;
    (void)tlast;
    u32 nblk = 0;
    while (st.chunk < a.chunks) {
        dedupe_block<TAG32, TAGIN>(a, sm, st, ph, tlast);          // leaves the next block in st
        __syncthreads();          // the table and the ticket word are free again
        DD_PHASE(6);
        nblk++;
    }
    if (a.dbg && threadIdx.x == 0) {
        for (int k = 0; k < 8; k++) a.dbg[(u64)blockIdx.x * 16 + k] = ph[k];
        a.dbg[(u64)blockIdx.x * 16 + 8] = nblk;
    }
}

// the words of the blocks, moved together and taken apart: block v's words -> keys / counts [incl[v] - nwords[v], incl[v])
// out_m (or null): beside them the mirrored words (rc(key) << pack | count), already grouped by their low block bits -- block v of
// the list IS group rc(v) of the mirror list (the first bases of a k-mer are the last of its reverse complement), and minc holds
// the groups' inclusive ends: the first stage of the mirror sort comes for free with the copy that is made anyway.

__global__ __launch_bounds__(256) void dedupe_unpack_kernel(const u64* __restrict__ in, const u64* __restrict__ cuts, const u64* __restrict__ incl,
                                                            const u64* __restrict__ nwords, u32 chunks, int pack, u64* __restrict__ out_k,
                                                            u32* __restrict__ out_c, u64* __restrict__ out_m, const u64* __restrict__ minc,
                                                            int K, int gbases, MirrorHist mh, const u64* __restrict__ place24, int packed_out) {
    __shared__ u32 bins[4 * 512];          // the digit histograms of the mirror sort's passes: it reads every word anyway
    const bool hist = out_m && mh.passes > 0;
    if (hist) {
        for (int q = threadIdx.x; q < 4 * 512; q += blockDim.x) bins[q] = 0;
        __syncthreads();
    }
    const u64 maxc = (1ull << pack) - 1;
    for (u32 v = blockIdx.x; v < chunks; v += gridDim.x) {
        const u64 cnt = nwords[v];
        const u64 dst0 = incl[v] - cnt;
        const u64* src = in + cuts[v];
        const u64 mdst = (out_m && !place24) ? minc[(u32)revcomp(gbases, (u64)v)] - cnt : 0;
        const int t6 = 2 * K - 2 * gbases - 6;          // where the 6 bits after the block bits sit in a key
        // four words of a thread in flight at a time (a block is ~3 K words: twelve rounds of one load each otherwise)
        for (u64 i0 = threadIdx.x; i0 < cnt; i0 += 4ull * blockDim.x) {
            u64 w4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u64 i = i0 + (u64)q * blockDim.x;
                w4[q] = i < cnt ? src[i] : 0;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u64 i = i0 + (u64)q * blockDim.x;
                if (i >= cnt) break;
                const u64 w = w4[q];
                if (packed_out) out_k[dst0 + i] = w;          // (the union reads the words as they are: 12 bytes less moved per entry)
                else { out_k[dst0 + i] = w >> pack; out_c[dst0 + i] = (u32)(w & maxc); }
                if (out_m) {
                    const u64 mw = (revcomp(K, w >> pack) << pack) | (w & maxc);
                    // place24: grouped by 6 more bits -- the block is sorted, so the words that share their next three bases are
                    // a run of it, and place24[v][those 6 bits] + i is the run's place in the group of the mirrored words
                    const u64 at = place24 ? place24[(u64)v * 64 + ((u32)(w >> (pack + t6)) & 63u)] + i : mdst + i;
                    out_m[at] = mw;
                    if (hist) {
#pragma unroll
                        for (int p = 0; p < 4; p++)
                            if (p < mh.passes) atomicAdd(&bins[p * 512 + ((u32)(mw >> mh.shift[p]) & ((1u << mh.bits[p]) - 1u))], 1u);
                    }
                }
            }
        }
    }
    if (hist) {
        __syncthreads();
        for (int q = threadIdx.x; q < mh.passes * 512; q += blockDim.x)
            if (bins[q]) atomicAdd(&mh.raw[q], (u64)bins[q]);
    }
}

// msz24[g] = words of the run (block v, 6-bit start j) whose mirror image is group g = rc3(j) << (2 gbases) | rc(v)
__global__ void dedupe_mirror_sizes24_kernel(const u32* __restrict__ sub, u32 chunks, int gbases, u64* __restrict__ msz) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (u64)chunks * 64) return;
    const u32 v = (u32)revcomp(gbases, g & ((u64)chunks - 1)), j = (u32)revcomp(3, g >> (2 * gbases));
    msz[g] = sub[(u64)v * 64 + j];
}
// place24[v][j] = (start of group g(v, j) in the mirror list) - (start of the run inside block v): add the word's index in the block
__global__ void dedupe_mirror_place24_kernel(const u32* __restrict__ sub, const u64* __restrict__ minc24, u32 chunks, int gbases,
                                             u64* __restrict__ place) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;          // one wavefront per block: lane = j
    const u32 v = (u32)(t >> 6), j = (u32)(t & 63);
    if (v >= chunks) return;
    const u32 x = sub[(u64)v * 64 + j];
    const u32 before = wave_incl_scan_u32(x) - x;
    const u64 g = ((u64)revcomp(3, (u64)j) << (2 * gbases)) | revcomp(gbases, (u64)v);
    place[(u64)v * 64 + j] = minc24[g] - x - before;
}

// msz[g] = words of the block whose mirror image is group g
__global__ void dedupe_mirror_sizes_kernel(const u64* __restrict__ nwords, u32 chunks, int gbases, u64* __restrict__ msz) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < chunks) msz[g] = nwords[(u32)revcomp(gbases, (u64)g)];
}

// the counts that did not fit a word: found again by key in the sorted list
__global__ void dedupe_big_kernel(const u64* __restrict__ big, u32 n_big, const u64* __restrict__ k, u64 n, u32* __restrict__ c, u32* err) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_big) return;
    const u64 key = big[2 * (u64)t], cnt = big[2 * (u64)t + 1];
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo < n && k[lo] == key) c[lo] = (u32)cnt; else atomicOr(err, ZK_DERR_CAPACITY);
}

// ---------------------------------------------------------------------------------------
// The same pass as a persistent, two-stage pipeline (array source, keys only).
//
// Measured on the one-tile-per-workgroup kernel: a tile spends a third of its life in the look-back,
// and nearly all of that is polling words that its neighbours -- started within a microsecond of it --
// have stored but that are not visible yet.  Waiting cannot be made shorter, so it is filled: a
// workgroup ranks tile B and publishes B's counts BEFORE it resolves the offsets of tile A, which it
// ranked one iteration earlier and parked, already grouped by digit, in LDS.  By then A's neighbours
// published 10 us ago and the look-back finds everything in place.
//
// Progress: a ticket, once taken, is ranked and published without waiting for any other tile, so every
// count a look-back waits for is on its way.
// ---------------------------------------------------------------------------------------
#ifdef ZK_STAMPS
#define PSTAMP(t, k) do { if (a.dbg && threadIdx.x == 0) a.dbg[(u64)(t) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PSTAT(t) ((a.dbg2 && threadIdx.x == 0) ? a.dbg2 + (t) : nullptr)
#else
#define PSTAMP(t, k) do { } while (0)
#define PSTAT(t) nullptr
#endif
// Tile order of the pipeline.  Counter x hands out the tiles of the chunks x, x+nx, x+2nx, ... (a chunk = 2^glog
// consecutive tiles) in ascending order, and a workgroup asks the counter of the XCD it runs on: neighbouring
// tiles -- whose output runs are neighbours in memory, 16 keys = one 128-byte line per digit and tile -- are
// then written through the same L2 within microseconds of each other.
// A workgroup whose own counter has run out takes from the others, so every tile is handed out whatever
// the placement of the workgroups (the XCD id only steers, it is never relied on).
// Measured (10^9 keys): runs of 16 tiles 2.97 TB/s, of 64 tiles 2.70, one global order 3.07 -- the scanners
// hand out offsets in tile order, so an XCD working ahead of its turn only waits; OFF by default (nx = 1).
__device__ __forceinline__ u32 tile_of(u32 x, u32 k, u32 nx, u32 glog) {
    return ((((k >> glog) * nx) + x) << glog) | (k & ((1u << glog) - 1u));
}
__device__ __forceinline__ u32 steal_tile(const SortArgs& a, u32 x, u32 tiles) {
    for (u32 y = 1; y < a.nx; y++) {
        const u32 xx = (x + y) % a.nx;
        const u32 t = tile_of(xx, atomicAdd(a.xticket + 32 * xx, 1u), a.nx, a.glog);
        if (t < tiles) return t;
    }
    return 0xffffffffu;
}

template <class C, bool PAIRS = false>
struct PipeSmem {
    static constexpr bool IMG_FITS = sizeof(TileImage<C::TILE>) <= sizeof(u16) * C::NW * C::RADIX;
    // the image takes the whole counter area: room for record-aligned tiles, whose byte span exceeds TILE
    static constexpr int IMG_WORDS = (int)(sizeof(u16) * C::NW * C::RADIX / 8);     // per array (codes, valid)
    static constexpr int IMG_T = IMG_FITS ? (IMG_WORDS - 3) * 16 : 16;
    u64 exch[C::TILE];              // tile A, grouped by digit, until its offsets are known
    u32 exv[PAIRS ? C::TILE : 1];   // ... and its payloads
    union {
        u16 cnt[C::NW][C::RADIX];
        TileImage<PipeSmem::IMG_T> img;     // stream source: the 2-bit image of tile B, dead before cnt is zeroed
    };
    u32 digit_off[C::RADIX];
    u64 gbase[C::RADIX];
    u32 wsum[C::NW];
    u32 ticket;
    u32 total_live;
    u32 abort;                      // a wait for offsets gave up (ZK_DERR_SPIN_TIMEOUT): the workgroup stores nothing from then on
};

// VAR: 1 only names the instantiation = the upper-bit passes over collapsed lists of packed words, so that a profiler lists them
// apart from the dominant full-size passes (same code); 2 = the pass stores the keys' low 32 bits only (SortArgs::tags_out).
// 3 = 2, and the keys of a tile take their places in their digit's run from a returning LDS add instead of the ballots: the pass
// before the block dedupe is the last one, nobody looks at the order inside a block -- all that must survive is that a block's keys
// stay together, i.e. the order of the BUCKETS of the pass before.  A tile that lies inside one such bucket (all but one in 1500 at
// config 2's size) has nothing to keep; a tile in which a bucket begins (SortArgs::straddle) is ranked by the ballots as before.
// 4 = the same for whole keys: the FIRST array pass after a pass over the stream, whatever the plan -- the stream pass has no order of
// its own, so all the second pass must keep is the first one's buckets.
template <class C, int SRC, int VAR = 0>
__global__ __launch_bounds__(C::BLOCK, C::WPE) void pass_pipe_kernel(SortArgs a, u32 tiles) {
    constexpr int BLOCK = C::BLOCK, ITEMS = C::ITEMS, RADIX = C::RADIX, NW = C::NW, DPT = C::DPT, TILE = C::TILE;
    constexpr bool PAIRS = VAR == 5 || VAR == 6;          // a 32-bit payload travels with every key (SortArgs::vin / vout)
    constexpr bool ATOM = VAR == 3 || VAR == 4 || VAR == 6, TAGS = VAR == 2 || VAR == 3;
    static_assert(!PAIRS || SRC == SRC_ARRAY, "pairs come from arrays");
    static_assert(!ATOM || (SRC == SRC_ARRAY && NW >= 2), "the counters of the adds are the first two rows of the per-wave counters");
    static_assert(C::ROUNDS == 1, "the pipeline parks a whole tile in LDS");
    constexpr int NS = RADIX / 64;          // scanner workgroups: 64 digits each
    __shared__ PipeSmem<C, VAR == 5 || VAR == 6> sm;
    static_assert(SRC == SRC_ARRAY || PipeSmem<C>::IMG_FITS, "the tile image lives in the counter area");
    static_assert(SRC == SRC_ARRAY || (ITEMS == 16 && PipeSmem<C>::IMG_WORDS <= 2 * BLOCK), "stream source: 16 consecutive windows per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 dmask = (1u << a.bits) - 1u;
    u16* mycnt = sm.cnt[wave];

    // ---- the first NS tickets are the scanners ------------------------------------------------------
    // A scanner workgroup owns 64 digits (one per lane).  Its NW waves take the tiles in batches of U, wave w
    // the batches w, w+NW, ...: read the U 16-bit counts (independent loads), add them up, take the running
    // total from the wave before through one LDS word per digit (sequence number | total), pass it on, and
    // only then write the U exclusive offsets (64-bit, epoch-tagged).  The chain between batches is an LDS
    // round trip; the slow parts -- global loads and store acknowledgements -- overlap across the waves.
    // Scanners hold the oldest tickets of the launch, hence they are resident, and every count they wait
    // for is published by a workgroup that holds a ticket.
    if (tid == 0) sm.abort = 0;
    u32* cnt32 = reinterpret_cast<u32*>(&sm.cnt[0][0]);          // VAR 3: one counter per digit for the whole workgroup
    if (ATOM && tid < RADIX) cnt32[tid] = 0;
    bool dirty = false;          // ... a tile ranked by the ballots has left its per-wave counts there
    const u32 first = take_ticket(a.ticket, &sm.ticket) - a.ticket_base;
    if (first < (u32)NS) {
        constexpr int U = 32;
        __builtin_amdgcn_s_setprio(3);              // every tile waits for these waves: issue them first
        u64* carry = sm.exch;                       // [64]: batch number << 40 | running total
        if (tid < 64) carry[tid] = 0;
        __syncthreads();
        // Buffer addressing: the batch's base in SGPRs, the lane's offset in one VGPR, the tile's offset as a scalar --
        // no 64-bit address per load or store in the vector registers (64 of each would not fit).  The buffer range
        // check is not relied on: every access is predicated on the tile index.
        const u64 tag = st_pack(ZK_ST_INCLUSIVE, a.epoch, 0);
        const u32 batches = (tiles + U - 1) / U;
        for (u32 b = (u32)wave; b < batches; b += NW) {
            const u32 t0 = b * U;
            const u32 nt = (tiles - t0 < (u32)U) ? tiles - t0 : (u32)U;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (u64)t0 * RADIX + (u64)first * 64), 0, 0x7fffffff, 0x00020000);
            __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(a.status + (u64)t0 * RADIX + (u64)first * 64), 0, 0x7fffffff, 0x00020000);
            u16 x[U];
#pragma unroll
            for (int k = 0; k < U; k++)
                x[k] = ((u32)k < nt) ? (u16)__builtin_amdgcn_raw_buffer_load_b16(rs, 2 * lane, k * RADIX * 2, 16) : (u16)0x8000u;
            // counts that were not published yet are asked for again TOGETHER: one round trip per retry, not one per
            // missing word (the scanner runs close behind the tiles, so first reads often come back empty)
            for (int spins = 0;; spins++) {
                u32 all = 0x8000u;
#pragma unroll
                for (int k = 0; k < U; k++) all &= x[k];
                if (all & 0x8000u) break;
                if (spins > ZK_SPIN_LIMIT) { atomicOr(a.err, ZK_DERR_SPIN_TIMEOUT | (8u << 8)); break; }
                __builtin_amdgcn_s_sleep(2);
#pragma unroll
                for (int k = 0; k < U; k++)
                    if (!(x[k] & 0x8000u)) x[k] = (u16)__builtin_amdgcn_raw_buffer_load_b16(rs, 2 * lane, k * RADIX * 2, 16);
            }
            u32 sum = 0;
#pragma unroll
            for (int k = 0; k < U; k++) {
                x[k] &= 0x7fffu;
                sum += x[k];
            }
            // hand-off: wait for batch b-1's total
            volatile u64* cw = carry + lane;
            u64 cv = *cw;
            int spins = 0;
            while ((u32)(cv >> 40) != b) {
                if (++spins > ZK_SPIN_LIMIT) { atomicOr(a.err, ZK_DERR_SPIN_TIMEOUT | (16u << 8)); break; }
                __builtin_amdgcn_s_sleep(1);
                cv = *cw;
            }
            u64 run = cv & ((1ull << 40) - 1);
            *cw = ((u64)(b + 1) << 40) | (run + sum);
            run |= tag;                              // totals stay below 2^40, the tag sits above: plain adds keep it
            // A write-through store is one fabric write per lane whatever its width: even lanes store their own
            // offset and their odd neighbour's as 16 bytes -- half the fabric writes of 64 eight-byte stores.
#pragma unroll
            for (int k = 0; k < U; k++) {
                const u32 nlo = (u32)__shfl_down((u32)run, 1, 64), nhi = (u32)__shfl_down((u32)(run >> 32), 1, 64);
                if ((u32)k < nt && !(lane & 1)) {
                    u32x4 w = {(u32)run, (u32)(run >> 32), nlo, nhi};
                    __builtin_amdgcn_raw_buffer_store_b128(w, rd, 8 * lane, k * RADIX * 8, 16);
                }
                run += x[k];
            }
        }
        return;
    }
    bool have = false;
    u32 tileA = 0, totalA = 0;
    u32 tcA[DPT], dexA[DPT];
#pragma unroll
    for (int j = 0; j < DPT; j++) { tcA[j] = 0; dexA[j] = 0; }

    // Software pipeline over the tiles: C's number is asked for while B is ranked, an iteration early.
    const u32 myx = (a.nx > 1) ? xcc_id() % a.nx : 0u;
    if (tid == 0) {
        u32 t = tile_of(myx, atomicAdd(a.xticket + 32 * myx, 1u), a.nx, a.glog);
        if (t >= tiles) t = steal_tile(a, myx, tiles);
        sm.ticket = t;
    }
    __syncthreads();
    u32 tB = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
    bool vB = tB < tiles;
    __syncthreads();          // sm.ticket is rewritten below
    NextTile<C, SRC> nx;
    u32 pending = 0;
    auto issue_loads = [&](u32 t) {
        if constexpr (SRC == SRC_ARRAY) nx.template issue<PAIRS>(a, t, tid, wave, lane);
        else nx.issue(a, t, tid, wave, lane);
    };
    if (vB) {
        if (tid == 0) pending = atomicAdd(a.xticket + 32 * myx, 1u);     // for the tile AFTER the one being loaded
        issue_loads(tB);
    }

    for (;;) {
        if (!vB && !have) break;
        if (vB) PSTAMP(tB, 0);
        // opaque per iteration: the 64-bit addresses built from the thread index (status row, histogram row, key base) are
        // then recomputed where they are used instead of living -- or being spilled -- across the whole tile loop
        u32 tid_o = (u32)tid;
        asm volatile("" : "+v"(tid_o));
        u64 key[ITEMS];
        u32 val[PAIRS ? ITEMS : 1];
        u32 rank2[ITEMS / 2];          // two 16-bit ranks per register
        u32 live = 0;
        u32 tC = 0;
        bool vC = false;
        bool atomB = false;          // VAR 3: tile B lies inside one bucket of the pass before: places from LDS adds
        bool early = false;          // the next tile's keys were asked for before tile A's stores
        u32 tcB[DPT], dexB[DPT];
        u32 incB = 0, tsumB = 0;
        u64 rowA[DPT];
#pragma unroll
        for (int j = 0; j < DPT; j++) { tcB[j] = 0; dexB[j] = 0; rowA[j] = 0; }

        if (vB) {
            if constexpr (SRC == SRC_ARRAY) {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) key[i] = nx.key[i];
                if constexpr (PAIRS) {
#pragma unroll
                    for (int i = 0; i < ITEMS; i++) val[i] = nx.val[i];
                    if (a.mirror_K > 0) {          // (the strand rebuild's first pass: the keys are mirrored as they are loaded)
#pragma unroll
                        for (int i = 0; i < ITEMS; i++) key[i] = revcomp(a.mirror_K, key[i]);
                    }
                }
                live = nx.live;
            } else {
                __syncthreads();      // the waves that parked the previous tile are done with the counters (same LDS)
                const u32 nch = a.rec ? (a.rpt * a.rec) / 16 + 3 : (u32)TileImage<TILE>::NCH;
                u32 cc, vv;
                if ((u32)tid < nch) {
                    encode_words16(nx.q0, cc, vv);
                    sm.img.codes[tid] = cc; sm.img.valid[tid] = vv;
                }
                if ((u32)tid + BLOCK < nch) {
                    encode_words16(nx.q1, cc, vv);
                    sm.img.codes[tid + BLOCK] = cc; sm.img.valid[tid + BLOCK] = vv;
                }
                __syncthreads();
                u64 xs[16], xr[16];
                // One code path for both tile shapes (two would double the registers): the thread's first window is at
                // tile position p0 -- 16 * tid for tiles of TILE positions; for record-aligned tiles thread = (record r
                // of the tile, chunk j of its windows), so that no slot is spent on the K windows per record that
                // run into the separator.
                // (from the opaque thread index and by a multiply: cheap enough to redo per tile, so that p0 and lim are not kept
                // -- or spilled -- across the tile loop)
                u32 p0 = 16u * tid_o, lim = 0xffffu;
                if (a.rec) {
                    const u32 r = (u32)(((u64)tid_o * a.cpr_inv) >> 32), j = tid_o - r * a.cpr;
                    p0 = r * a.rec + 16u * j;
                    const int left = (int)a.wpr - 16 * (int)j;              // windows of the record from this chunk on
                    lim = (r < a.rpt) ? ((left >= 16) ? 0xffffu : ((1u << (left > 0 ? left : 0)) - 1u)) : 0u;
                    if (r >= a.rpt) p0 = 0;
                }
                live = windows16_at(sm.img, (int)p0, a.K, xs, xr) & lim;
#pragma unroll
                for (int i = 0; i < ITEMS; i++) {
                    const u64 x = xs[i & 15], xb = xr[i & 15];
                    key[i] = (a.mode == ZK_KEYS_CANONICAL) ? (x < xb ? x : xb) : x;
                }
                __syncthreads();      // the image is dead: its space becomes the counters
            }
            // (Pass 0 has no order to keep, so its ranks could come from a returning LDS add per key instead of the ballots: measured,
            // 32.7 vs 30.9 ms.)
            if constexpr (ATOM) atomB = !((a.straddle[tB >> 5] >> (tB & 31u)) & 1u);
            if (ATOM && atomB) {
                if (dirty) {          // (the tile before was ranked by the ballots and is being parked with those counts)
                    __syncthreads();
                    if (tid < RADIX) cnt32[tid] = 0;
                    __syncthreads();
                    dirty = false;
                }
                PSTAMP(tB, 1);
#pragma unroll
                for (int i = 0; i < ITEMS; i++) {
                    const bool lv = (live >> i) & 1u;
                    const u32 d = (u32)(key[i] >> a.shift) & dmask;
                    const u32 r = lv ? atomicAdd(&cnt32[d], 1u) : 0u;
                    if (i & 1) rank2[i / 2] |= r << 16; else rank2[i / 2] = r;
                }
            } else {
            if constexpr (ATOM) dirty = true;
            // this wave's counters: private to the wave until the scan, so no barrier after zeroing them
            {
                u32 z = 0;          // made here, not kept: a zero quad held (or spilled) across the tile loop costs four registers
                asm volatile("" : "+v"(z));
                for (int q = lane; q < RADIX / 8; q += 64) reinterpret_cast<uint4*>(mycnt)[q] = make_uint4(z, z, z, z);
            }
#ifdef ZK_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PSTAMP(tB, 1);
            // ---- rank inside the wave (see pass_kernel) ------------------------------------------
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                const bool lv = (live >> i) & 1u;
                const u32 d = (u32)(key[i] >> a.shift) & dmask;
                u32 plo, phi;
                match_digit<C::RBITS>(d, __ballot(lv), plo, phi);
                const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                const u32 npeer = (u32)__popc(plo) + (u32)__popc(phi);
                const u32 pre = lv ? (u32)mycnt[d] : 0u;
                if (i & 1) rank2[i / 2] |= (pre + below) << 16; else rank2[i / 2] = pre + below;
                if (lv && below == npeer - 1) mycnt[d] = (u16)(pre + npeer);
            }
            }
            PSTAMP(tB, 2);
            // the digits are recomputed when the tile is parked (two instructions each) instead of sixteen more registers
            // staying live across tile A's stores -- room for the next tile's keys to be in flight by then
            if constexpr (SRC == SRC_ARRAY) {
#pragma unroll
                for (int i = 0; i < ITEMS; i++) asm volatile("" : "+v"(key[i]));
            }
            // the atomic was issued before this tile's key loads, so it has returned by now (no extra wait)
            if (tid == 0) {
                u32 t = tile_of(myx, pending, a.nx, a.glog);
                if (t >= tiles) t = steal_tile(a, myx, tiles);
                sm.ticket = t;
            }
            if (have) {
#pragma unroll
                for (int j = 0; j < DPT; j++) {
                    const int d = (int)tid_o * DPT + j;
                    if (d < RADIX) rowA[j] = ld_agent(a.status + (u64)tileA * RADIX + d);   // consumed after B is published
                }
            }
            __syncthreads();
            tC = (u32)__builtin_amdgcn_readfirstlane((int)sm.ticket);
            vC = tC < tiles;
            if constexpr (SRC == SRC_ARRAY) {
                early = vC && (u64)(tC + 1) * TILE <= a.n;          // whole tiles only: one load path, no join
                if (early) {
                    if (tid == 0) pending = atomicAdd(a.xticket + 32 * myx, 1u);
                    const u64* p = a.kin + (u64)tC * TILE + (u64)(wave * (64 * ITEMS) + (int)(tid_o & 63u));
#pragma unroll
                    for (int i = 0; i < ITEMS; i++) nx.key[i] = p[i * 64];
                    if constexpr (PAIRS) {
                        const u32* q = a.vin + (u64)tC * TILE + (u64)(wave * (64 * ITEMS) + (int)(tid_o & 63u));
#pragma unroll
                        for (int i = 0; i < ITEMS; i++) nx.val[i] = q[i * 64];
                    }
                    nx.live = (1u << ITEMS) - 1u;
                }
            }          // (stream source: its next tile is only 16-32 bytes per thread; asking for it here as well was slower, 31.5 vs 30.8 ms)
            // ---- per digit: scan over the waves, publish the tile's count at once -------------------
            u32 tsum = 0;
#pragma unroll
            for (int j = 0; j < DPT; j++) {
                const int d = tid * DPT + j;
                u32 acc = 0;
                if (ATOM && atomB) {
                    if (d < RADIX) { acc = cnt32[d]; cnt32[d] = 0; }          // (zero again for the next tile: nobody reads it before the barriers below)
                } else
                if (d < RADIX) {
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const u32 t = sm.cnt[w][d];
                        sm.cnt[w][d] = (u16)acc;
                        acc += t;
                    }
                }
                static_assert(DPT == 1 && BLOCK >= RADIX && RADIX % 64 == 0, "one digit per thread in the pipeline");
                if (wave * 64 < RADIX) publish_counts16(a.part + (u64)tB * RADIX, d, acc, lane);     // whole waves
                tcB[j] = acc;
                tsum += acc;
            }
            incB = wave_incl_scan_u32(tsum);
            tsumB = tsum;
            if (lane == 63) sm.wsum[wave] = incB;     // read after the next barrier
        }

        if (vB) PSTAMP(tB, 4);
        if (have) {
            // ---- tile A: offsets, then out of LDS ------------------------------------------------
#pragma unroll
            for (int j = 0; j < DPT; j++) {
                const int d = (int)tid_o * DPT + j;
                if (d < RADIX) {
                    const u64* q = a.status + (u64)tileA * RADIX + d;
                    u64 w = vB ? rowA[j] : ld_agent(q);
                    int spins = 0;
#ifdef ZK_PHASES
                    if (a.dbg_local) w = st_pack(ZK_ST_INCLUSIVE, a.epoch, 0);
#endif
                    while (st_state(w, a.epoch) == 0) {
                        if (++spins > ZK_SPIN_LIMIT) { atomicOr(a.err, ZK_DERR_SPIN_TIMEOUT | (32u << 8)); sm.abort = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                        w = ld_agent(q);
                    }
#ifdef ZK_STAMPS
                    if (PSTAT(tileA)) *PSTAT(tileA) = (1ull << 32) | (u32)spins;
#endif
                    sm.gbase[d] = a.ghist[d] + (w & ZK_ST_VALUE_MASK) - dexA[j];
#ifdef ZK_PHASES
                    if (a.dbg_local) sm.gbase[d] = (u64)tileA * TILE;
#endif
                }
            }
            PSTAMP(tileA, 5);
#ifdef ZK_STAMPS
            if (a.dbg && lane == 0) a.dbg[9ull * tiles + 8ull * tileA + wave] = __builtin_amdgcn_s_memtime();   // per wave: offsets known
#endif
        }
        __syncthreads();          // A's offsets and B's per-wave digit sums are visible
        if (vB) {
            // ---- exclusive scan over the digits of B ------------------------------------------------
            u32 woff = 0;
            for (int w = 0; w < wave; w++) woff += sm.wsum[w];
            u32 run = woff + incB - tsumB;
            if (tid == BLOCK - 1) sm.total_live = woff + incB;
#pragma unroll
            for (int j = 0; j < DPT; j++) {
                const int d = tid * DPT + j;
                dexB[j] = run;
                if (d < RADIX) sm.digit_off[d] = run;
                run += tcB[j];
            }
        }
        if (have && sm.abort) totalA = 0;          // offsets that never came mean nothing: no store is made with them (the launch reports the error)
        if (have) {
            PSTAMP(tileA, 6);
            // The slot index is made opaque: otherwise the compiler precomputes the sixteen `kout + slot` addresses
            // outside the tile loop, spills them, and every store then waits (vmcnt is in order) for the reload
            // of its address and with it for the store before it -- sixteen serial round trips per tile.
            u32 slot0 = (u32)tid;
            asm volatile("" : "+v"(slot0));
            // eight slots at a time: the LDS reads of a group are independent of each other (slots past the live
            // count hold stale but readable keys), only the global store is predicated
            constexpr int G = (SRC == SRC_ARRAY) ? 2 : 4;
#pragma unroll
            for (int i0 = 0; i0 < ITEMS; i0 += G) {
                u64 kk[G], pos[G];
#pragma unroll
                for (int g = 0; g < G; g++) kk[g] = sm.exch[slot0 + (i0 + g) * BLOCK];
#pragma unroll
                for (int g = 0; g < G; g++) pos[g] = sm.gbase[(u32)(kk[g] >> a.shift) & dmask] + (slot0 + (i0 + g) * BLOCK);
#pragma unroll
                for (int g = 0; g < G; g++)
                    if (slot0 + (i0 + g) * BLOCK < totalA) {
                        if constexpr (TAGS) reinterpret_cast<u32*>(a.kout)[pos[g]] = (u32)kk[g];
                        else a.kout[pos[g]] = kk[g];
                        if constexpr (PAIRS) a.vout[pos[g]] = sm.exv[slot0 + (i0 + g) * BLOCK];
                    }
            }
        }
#ifdef ZK_STAMPS
        if (have) { PSTAMP(tileA, 7); }
#endif
        if (!vB) break;
        __syncthreads();      // exch is free again; digit_off / total_live of B are visible
        // ---- park B, grouped by digit ------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const u32 d = (u32)(key[i] >> a.shift) & dmask;
            const u32 inwave = (ATOM && atomB) ? 0u : (u32)sm.cnt[wave][d];
            const u32 at = sm.digit_off[d] + inwave + ((rank2[i / 2] >> (16 * (i & 1))) & 0xffffu);
            if ((live >> i) & 1u) {
                sm.exch[at] = key[i];
                if constexpr (PAIRS) sm.exv[at] = val[i];
            }
        }
        tileA = tB;
        totalA = sm.total_live;
#pragma unroll
        for (int j = 0; j < DPT; j++) { tcA[j] = tcB[j]; dexA[j] = dexB[j]; }
        PSTAMP(tB, 3);          // B is parked
        have = true;
        tB = tC;
        vB = vC;
        // Array source: a whole tile C was asked for right after B was ranked (above) -- that fits the 128 registers only
        // because the digits, the thread-index addresses and the zero quad are NOT kept across the iteration (the opaque
        // copies above; before, 16 early rows spilled and 4 were slower than none).  Worth 2 % of the pass.
        // Stream source and the cut last tile: the loads start here.
        if (vB && !early) {
            if (tid == 0) pending = atomicAdd(a.xticket + 32 * myx, 1u);
            if constexpr (SRC == SRC_ARRAY) {
                // only the cut last tile comes here: per-key bounds
                const u64 base = (u64)tB * TILE + (u64)wave * (64 * ITEMS) + lane;
                nx.live = 0;
#pragma unroll
                for (int i = 0; i < ITEMS; i++) {
                    const u64 idx = base + (u64)i * 64;
                    const bool ok = idx < a.n;
                    nx.key[i] = ok ? a.kin[idx] : 0ull;
                    if constexpr (PAIRS) nx.val[i] = ok ? a.vin[idx] : 0u;
                    nx.live |= (ok ? 1u : 0u) << i;
                }
            } else {
                issue_loads(tB);
            }
        }
    }
}

// starts[b] = where bucket b of the pass before begins in this pass's input: the tile it begins in (unless it begins at the tile's
// first key) holds keys of two buckets or more
__global__ void straddle_kernel(const u64* __restrict__ starts, u32 buckets, u32 tile, u32* __restrict__ bitmap) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 || b >= buckets) return;
    const u64 pos = starts[b];
    if (pos % tile) atomicOr(&bitmap[(pos / tile) >> 5], 1u << ((pos / tile) & 31u));
}

// Where do the blocks of the block dedupe start when the last pass has written tags (no key to search for)?  Block (dh, v) --
// dh the last pass's digit, v the digit of the pass before it -- starts at
//     (keys with a smaller dh) + (keys with this dh in the tiles before the one where bucket v of the pass's input begins)
//                              + (keys with this dh in that tile before bucket v)
// the first from the pass's histogram, the second from the offsets its scanners published per tile (still in the status words),
// the third counted here from the pass's input, a tile's worth per v.  One workgroup per v.
__global__ __launch_bounds__(512) void tag_cuts_kernel(const u64* __restrict__ kin, u64 n, const u64* __restrict__ ghist_in, const u64* __restrict__ ghist_d,
                                                       const u64* __restrict__ status, u32 stride, u32 tile, u32 tiles, int shift, int bits_d, int bits_in,
                                                       u64* __restrict__ cuts) {
    __shared__ u32 corr[1024];
    const u32 v = blockIdx.x, Rd = 1u << bits_d, Rin = 1u << bits_in;
    for (u32 d = threadIdx.x; d < Rd; d += blockDim.x) corr[d] = 0;
    __syncthreads();
    const u64 boundary = ghist_in[v];
    const u64 tv = boundary / tile;
    for (u64 i = tv * tile + threadIdx.x; i < boundary; i += blockDim.x) atomicAdd(&corr[(u32)(kin[i] >> shift) & (Rd - 1u)], 1u);
    __syncthreads();
    for (u32 d = threadIdx.x; d < Rd; d += blockDim.x) {
        u64 before;
        if (tv < tiles) before = status[tv * stride + d] & ZK_ST_VALUE_MASK;
        else before = (d + 1 < Rd ? ghist_d[d + 1] : n) - ghist_d[d];          // the bucket begins at the very end: every key of dh lies before it
        cuts[(u64)d * Rin + v] = ghist_d[d] + before + corr[d];
    }
    if (v == 0 && threadIdx.x == 0) cuts[(u64)Rd * Rin] = n;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int launch_wide_pass(zk_ctx* c, const SortArgs& a);
constexpr u64 WIDE_TILES_MAX_KEYS = ~0ull;          // (round 4: no limit any more, see V6)

template <class C>
struct Sorter {
    static u32 tiles_for(const SortArgs& a, int src) {
        if (src == SRC_ARRAY) return (u32)div_up(a.n, C::TILE);
        if (a.rec) return (u32)div_up(a.n_bytes / a.rec, a.rpt);           // record-aligned tiles (pipeline pass 0 only)
        const u64 pos = (a.mode == ZK_KEYS_BOTH) ? C::TILE / 2 : C::TILE;
        return (u32)div_up(a.n_bytes, pos);
    }

    template <int SRC, bool PAIRS>
    static int launch_pass(zk_ctx* c, SortArgs a) {
        const u32 tiles = tiles_for(a, SRC);
        if (tiles == 0) return ZK_OK;
        if (C::SEG > 0) {
            ZK_TRY(lookback_begin(c, ((uint64_t)tiles / C::SEG + 2) * C::RADIX, tiles, &a.epoch, &a.ticket_base));
            ZK_TRY(part16_begin(c, (uint64_t)tiles * C::RADIX, &a.part));
        } else {
            ZK_TRY(lookback_begin(c, (uint64_t)tiles * C::RADIX, tiles, &a.epoch, &a.ticket_base));
        }
        a.status = c->status;
        a.ticket = c->d_ticket;
        a.err = c->d_err;
        a.dbg = c->dbg;
        a.dbg2 = c->dbg ? c->dbg + 8ull * tiles : nullptr;
        prof_begin(c, SRC == SRC_STREAM ? ZK_PROF_PASS_STREAM : (PAIRS ? ZK_PROF_PASS_PAIRS : ZK_PROF_PASS_KEYS),
                   SRC == SRC_STREAM ? a.n_bytes + 8 * a.n : (PAIRS ? 24 : 16) * a.n);
        hipLaunchKernelGGL((pass_kernel<C, SRC, PAIRS>), dim3(tiles), dim3(C::BLOCK), 0, c->stream, a);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
        return ZK_OK;
    }

    // persistent pipelined pass (keys only)
    template <int SRC>
    static int launch_pipe(zk_ctx* c, SortArgs a) {
      if constexpr (C::PIPE) {
        const u32 tiles = tiles_for(a, SRC);
        if (tiles == 0) return ZK_OK;
        u32 grid = (u32)c->num_cus * (sizeof(PipeSmem<C>) > 80 * 1024 ? 1 : 2);          // what fits a CU's 160 KB of LDS
        if (grid > tiles) grid = tiles;
        grid += C::RADIX / 64;          // the scanner workgroups
        ZK_TRY(lookback_begin(c, (uint64_t)tiles * C::RADIX, grid, &a.epoch, &a.ticket_base));   // one role ticket per workgroup
        ZK_TRY(part16_begin(c, (uint64_t)tiles * C::RADIX, &a.part));
        ZK_HIP(c, hipMemsetAsync(c->d_xticket, 0, 8 * 32 * sizeof(u32), c->stream));
        a.xticket = c->d_xticket;
        a.nx = (c->xcd_group > 0 && c->num_xcd == 8) ? 8u : 1u;
        a.glog = 0;
        while ((1 << (a.glog + 1)) <= c->xcd_group) a.glog++;
        a.status = c->status;
        a.ticket = c->d_ticket;
        a.err = c->d_err;
        a.dbg = c->dbg;
        a.dbg2 = c->dbg ? c->dbg + 8ull * tiles : nullptr;
        prof_begin(c, SRC == SRC_STREAM ? ZK_PROF_PASS_STREAM : (a.prof_tag ? a.prof_tag : ZK_PROF_PASS_KEYS),
                   SRC == SRC_STREAM ? a.n_bytes + 8 * a.n : (a.tags_out ? 12 : 16) * a.n);
        if (SRC == SRC_ARRAY && a.tags_out && a.straddle)
            hipLaunchKernelGGL((pass_pipe_kernel<C, SRC_ARRAY, 3>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        else if (SRC == SRC_ARRAY && a.straddle)
            hipLaunchKernelGGL((pass_pipe_kernel<C, SRC_ARRAY, 4>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        else if (SRC == SRC_ARRAY && a.tags_out)
            hipLaunchKernelGGL((pass_pipe_kernel<C, SRC_ARRAY, 2>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        else if (SRC == SRC_ARRAY && a.prof_tag == ZK_PROF_PASS_PACKED)
            hipLaunchKernelGGL((pass_pipe_kernel<C, SRC, 1>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        else
            hipLaunchKernelGGL((pass_pipe_kernel<C, SRC, 0>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
      }
      return ZK_OK;
    }
    // the same pipeline with a 32-bit payload (VAR 5; 6 = places from LDS adds where a.straddle allows)
    static constexpr bool PIPE_PAIRS = C::PIPE && C::ITEMS <= 8;          // (a tile of pairs is 12 bytes an entry: 8 K of them)
    static int launch_pipe_pairs(zk_ctx* c, SortArgs a) {
      if constexpr (PIPE_PAIRS) {
        const u32 tiles = (u32)div_up(a.n, C::TILE);
        if (tiles == 0) return ZK_OK;
        u32 grid = (u32)c->num_cus * (sizeof(PipeSmem<C, true>) > 80 * 1024 ? 1 : 2);
        if (grid > tiles) grid = tiles;
        grid += C::RADIX / 64;          // the scanner workgroups
        ZK_TRY(lookback_begin(c, (uint64_t)tiles * C::RADIX, grid, &a.epoch, &a.ticket_base));
        ZK_TRY(part16_begin(c, (uint64_t)tiles * C::RADIX, &a.part));
        ZK_HIP(c, hipMemsetAsync(c->d_xticket, 0, 8 * 32 * sizeof(u32), c->stream));
        a.xticket = c->d_xticket;
        a.nx = 1u;
        a.glog = 0;
        a.status = c->status;
        a.ticket = c->d_ticket;
        a.err = c->d_err;
        a.dbg = nullptr; a.dbg2 = nullptr;
        prof_begin(c, ZK_PROF_PASS_PAIRS, 24 * a.n);
        if (a.straddle) hipLaunchKernelGGL((pass_pipe_kernel<C, SRC_ARRAY, 6>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        else hipLaunchKernelGGL((pass_pipe_kernel<C, SRC_ARRAY, 5>), dim3(grid), dim3(C::BLOCK), 0, c->stream, a, tiles);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
      }
      return ZK_OK;
    }
    // keys per tile of the geometry launch_keys_pass uses for this many keys
    static u32 keys_pass_tile(zk_ctx* c, uint64_t n) {
        if constexpr (C::PIPE && C::RBITS == 9 && C::BLOCK == 512) { if (c->wide_tiles && n <= WIDE_TILES_MAX_KEYS) return 16384u; }
        return (u32)C::TILE;
    }
    static int launch_keys_pass(zk_ctx* c, const SortArgs& a) {
        // the default geometry hands its array passes to the 16 K-key tiles (same digits, same histograms) while the array is
        // not too big for them: see V6
        if constexpr (C::PIPE && C::RBITS == 9 && C::BLOCK == 512) { if (c->wide_tiles && a.n <= WIDE_TILES_MAX_KEYS) return launch_wide_pass(c, a); }
        if (C::PIPE) return launch_pipe<SRC_ARRAY>(c, a);
        return launch_pass<SRC_ARRAY, false>(c, a);
    }

    template <int SRC>
    static int launch_hist(zk_ctx* c, const SortArgs& src, const PassPlan& plan, u64* ghist, u64* acgt, u64* d_n,
                           u64* sample = nullptr, u32 sample_cap = 0, int sample_shift = 0, u64 sample_value = 0) {
        HistArgs h;
        h.sample = sample; h.sample_cap = sample_cap; h.sample_shift = sample_shift; h.sample_value = sample_value;
        h.sample_n = (u32*)(c->d_scalars + 23);
        if (sample) ZK_HIP(c, hipMemsetAsync(c->d_scalars + 23, 0, sizeof(u64), c->stream));
        h.src = src;
        h.plan = plan;
        h.ghist = ghist;
        h.acgt = acgt;
        h.tiles = tiles_for(src, SRC);
        h.rec_info = nullptr;
        ZK_HIP(c, hipMemsetAsync(ghist, 0, sizeof(u64) * MAX_PASSES * C::RADIX, c->stream));
        if (SRC == SRC_STREAM && src.n_bytes) {
            h.rec_info = c->d_scalars + 20;
            ZK_HIP(c, hipMemsetAsync(h.rec_info, 0, 3 * sizeof(u64), c->stream));
            hipLaunchKernelGGL(first_newline_kernel, dim3(1), dim3(256), 0, c->stream, src.stream, (u64)src.n_bytes, h.rec_info);
        }
        if (acgt) ZK_HIP(c, hipMemsetAsync(acgt, 0, sizeof(u64) * 4, c->stream));
        u32 grid = h.tiles < (u32)(c->num_cus * 8) ? h.tiles : (u32)(c->num_cus * 8);
        if (grid == 0) grid = 1;
        prof_begin(c, SRC == SRC_STREAM ? ZK_PROF_HIST_STREAM : ZK_PROF_HIST_ARRAY, SRC == SRC_STREAM ? src.n_bytes : 8 * src.n);
        if constexpr (SRC == SRC_STREAM && C::ITEMS == 16 && C::BLOCK <= 512) {
            if (src.mode != ZK_KEYS_BOTH) hipLaunchKernelGGL((hist_kernel<C, SRC, true>), dim3(grid), dim3(C::BLOCK), 0, c->stream, h);
            else hipLaunchKernelGGL((hist_kernel<C, SRC, false>), dim3(grid), dim3(C::BLOCK), 0, c->stream, h);
        } else {
            hipLaunchKernelGGL((hist_kernel<C, SRC, false>), dim3(grid), dim3(C::BLOCK), 0, c->stream, h);
        }
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
        hipLaunchKernelGGL(hist_scan_kernel, dim3(1), dim3(256), 0, c->stream, ghist, plan.passes, (int)C::RADIX, d_n);
        ZK_HIP(c, hipGetLastError());
        return ZK_OK;
    }

    // lo_bit > 0: only the bits [lo_bit, key_bits) are sorted (the input is already ordered by the bits below)
    // counted: [passes][RADIX] raw digit counts that the producer of `keys` took on the way (no histogram pass of its own)
    static int sort_keys(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, u64** result, int lo_bit = 0, int prof_tag = 0,
                         u64* counted = nullptr) {
        PassPlan plan = make_plan(key_bits - lo_bit, C::RBITS, lo_bit);
        u64* ghist = counted;
        if (!counted) ZK_TRY(arena_alloc(c, sizeof(u64) * MAX_PASSES * C::RADIX, (void**)&ghist));
        SortArgs a = {};
        a.kin = keys; a.n = n;
        a.prof_tag = prof_tag;      // the upper-bit passes over collapsed / packed lists are timed apart
        if (counted) {
            hipLaunchKernelGGL(hist_scan_kernel, dim3(1), dim3(256), 0, c->stream, ghist, plan.passes, (int)C::RADIX, c->d_scalars + 8);
            ZK_HIP(c, hipGetLastError());
        } else {
            ZK_TRY(launch_hist<SRC_ARRAY>(c, a, plan, ghist, nullptr, c->d_scalars + 8));
        }
        u64* in = keys; u64* out = alt;
        for (int p = 0; p < plan.passes; p++) {
            a.kin = in; a.kout = out; a.shift = plan.shift[p]; a.bits = plan.bits[p];
            a.ghist = ghist + p * C::RADIX;
            ZK_TRY(launch_keys_pass(c, a));
            u64* t = in; in = out; out = t;
        }
        *result = in;
        return ZK_OK;
    }

    // mirror_K > 0: sort (rc(src_k[i]), src_v[i]) instead, without writing the mirrored keys first: the histogram and
    // the first pass apply rc on load; src_k / src_v are only read, keys/alt/vals/valt are the two work buffers.
    // lo_bit > 0: only the bits [lo_bit, key_bits) are sorted (the input is already ordered by the bits below)
    // unordered: nobody needs pairs of equal keys to stay in their order (the mirrored keys of the strand rebuild are all different) and
    // the order the keys arrive in means nothing: the first pass takes its places from LDS adds in every tile, the second wherever a
    // tile lies inside one bucket of the first (pipeline geometries only)
    static int sort_pairs(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, u64** rk, u32** rv,
                          const u64* src_k = nullptr, const u32* src_v = nullptr, int mirror_K = 0, int lo_bit = 0, bool unordered = false) {
        PassPlan plan = make_plan(key_bits - lo_bit, C::RBITS, lo_bit);
        u64* ghist;
        ZK_TRY(arena_alloc(c, sizeof(u64) * MAX_PASSES * C::RADIX, (void**)&ghist));
        SortArgs a = {};
        a.kin = mirror_K ? src_k : keys; a.n = n; a.mirror_K = mirror_K;
        ZK_TRY(launch_hist<SRC_ARRAY>(c, a, plan, ghist, nullptr, c->d_scalars + 8));
        const u64* in = mirror_K ? src_k : keys; const u32* vi = mirror_K ? src_v : vals;
        u64* out = mirror_K ? keys : alt; u32* vo = mirror_K ? vals : valt;
        for (int p = 0; p < plan.passes; p++) {
            a.kin = in; a.kout = out; a.vin = vi; a.vout = vo; a.shift = plan.shift[p]; a.bits = plan.bits[p];
            a.ghist = ghist + p * C::RADIX;
            a.mirror_K = (p == 0) ? mirror_K : 0;
            if constexpr (PIPE_PAIRS) {
                a.straddle = nullptr;
                if (unordered && p <= 1) {
                    const u32 tiles = (u32)div_up(n, C::TILE);
                    u32* bm;
                    ZK_TRY(arena_alloc(c, sizeof(u32) * (tiles / 32 + 1), (void**)&bm));
                    ZK_HIP(c, hipMemsetAsync(bm, 0, sizeof(u32) * (tiles / 32 + 1), c->stream));
                    if (p == 1) {
                        hipLaunchKernelGGL(straddle_kernel, dim3((C::RADIX + 255) / 256), dim3(256), 0, c->stream, (const u64*)(ghist + (p - 1) * C::RADIX),
                                           1u << plan.bits[p - 1], (u32)C::TILE, bm);
                        ZK_HIP(c, hipGetLastError());
                    }
                    a.straddle = bm;
                }
                ZK_TRY(launch_pipe_pairs(c, a));
            } else
            ZK_TRY((launch_pass<SRC_ARRAY, true>(c, a)));
            in = out; vi = vo;
            out = (out == keys) ? alt : keys;
            vo = (vo == vals) ? valt : vals;
        }
        *rk = const_cast<u64*>(in); *rv = const_cast<u32*>(vi);
        return ZK_OK;
    }

    // Sort the k-mers of a base stream without ever storing them unsorted: histogram and first
    // pass read the stream, the remaining passes ping-pong between buf_a and buf_b.
    static int sort_stream(zk_ctx* c, const StreamSrc& src, u64* buf_a, u64* buf_b, uint64_t cap, uint64_t* n_keys,
                           uint64_t acgt[4], u64** result) {
        const int top = (src.hi_bit > 0 && src.hi_bit < 2 * src.K) ? src.hi_bit : 2 * src.K;     // sort the bits [lo_bit, top)
        PassPlan plan = make_plan(top - src.lo_bit, C::RBITS, src.lo_bit);
        u64* ghist;
        ZK_TRY(arena_alloc(c, sizeof(u64) * MAX_PASSES * C::RADIX, (void**)&ghist));
        SortArgs a = {};
        a.stream = src.stream; a.n_bytes = src.n_bytes; a.K = src.K; a.mode = src.mode;
        u64* d_acgt = c->d_scalars + 0;
        u64* d_n = c->d_scalars + 8;
        // the look before the sort (StreamSample): the set-aside keys go to the second sort buffer, which is idle until pass 1
        const u32 sample_cap = 1u << 20;
        const bool sampling = src.sample && cap >= 4ull * sample_cap && src.mode == ZK_KEYS_CANONICAL;
        // stream_pass.hip: the pass over static stream ranges needs the digit counts of pass 0 per range, from its own histogram kernel
        const bool ranged = c->stream_pass != 0;          // (ZK_KEYS_BOTH: the two strands of a range as two ranges)
        StreamRows srows;
        if (ranged) {
            u64* rec_info = c->d_scalars + 20;
            ZK_HIP(c, hipMemsetAsync(rec_info, 0, 4 * sizeof(u64), c->stream));          // [20..22] the records, [23] the sample counter
            hipLaunchKernelGGL(first_newline_kernel, dim3(1), dim3(256), 0, c->stream, src.stream, (u64)src.n_bytes, rec_info);
            ZK_TRY(stream_hist(c, src.stream, src.n_bytes, src.K, src.mode, plan, ghist, (u32)C::RADIX, d_acgt, d_n, rec_info,
                               sampling ? buf_b : nullptr, sample_cap, sampling ? src.sample->shift : 0, sampling ? src.sample->value : 0,
                               (u32*)(c->d_scalars + 23),
                               // the stream's image goes behind the set-aside keys in the second sort buffer (idle until pass 1)
                               (char*)buf_b + (sampling ? 24ull * sample_cap : 0), 8 * cap - (sampling ? 24ull * sample_cap : 0), &srows));
            hipLaunchKernelGGL(hist_scan_kernel, dim3(1), dim3(256), 0, c->stream, ghist, plan.passes, (int)C::RADIX, d_n);
            ZK_HIP(c, hipGetLastError());
        } else {
            ZK_TRY(launch_hist<SRC_STREAM>(c, a, plan, ghist, d_acgt, d_n, sampling ? buf_b : nullptr, sample_cap,
                                           sampling ? src.sample->shift : 0, sampling ? src.sample->value : 0));
        }
        // the number of live keys decides the grids of the array passes: one small readback
        ZK_HIP(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(u64) * 24, hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        const uint64_t n = c->h_scalars[8];
        if (acgt) for (int b = 0; b < 4; b++) acgt[b] = c->h_scalars[b];
#ifdef ZK_PHASES
        // measurement of the histogram kernel with fewer LDS adds than it needs (tools/p0_phases.py): its counts are wrong, nothing
        // may be sorted by them
        if (const char* e = getenv("ZK_HIST_ATOMICS")) if (atoi(e) != 2) { *n_keys = 0; return ZK_OK; }
#endif
        *n_keys = n;
        if (n > cap) return fail(c, ZK_ENOSPC, "sort buffers hold %llu keys, the stream has %llu", (unsigned long long)cap, (unsigned long long)n);
        if (n == 0) return ZK_OK;
        if (sampling) {
            uint64_t sn = c->h_scalars[23] & 0xffffffffull;
            if (sn > sample_cap) sn = sample_cap;
            src.sample->seen = sn;
            if (sn >= 4096) {          // enough to judge
                u64* res = nullptr;
                uint64_t distinct = 0;
                ZK_TRY(sort_keys(c, buf_b, buf_b + sample_cap, sn, 2 * src.K, &res, 0, ZK_PROF_SAMPLE));
                ZK_TRY(rle(c, res, sn, res, (u32*)(buf_b + 2ull * sample_cap), sn, &distinct));
                src.sample->distinct = distinct;
                const bool repeats = (double)distinct <= src.sample->max_ratio * (double)sn;
                if (repeats == src.sample->want_distinct) return 1;          // declined: nothing sorted
            }
        }
        a.kout = buf_a; a.shift = plan.shift[0]; a.bits = plan.bits[0]; a.ghist = ghist;
        a.n = n;
        if (ranged) {
            const uint64_t first_nl = c->h_scalars[20], nl = c->h_scalars[21], bad = c->h_scalars[22];
            const bool uniform = first_nl < 0x7fffffffull && src.n_bytes % (first_nl + 1) == 0 && bad == 0 && nl == src.n_bytes / (first_nl + 1);
            // The usual plan (two passes over the top bits, tags for the block dedupe): pass 0 leaves the keys as two arrays, low words and
            // next digits, 6 bytes a key, and the second pass is tag_pass.hip's count / scan / scatter over static segments -- done here.
            const uint64_t n_pad = ((n + 63) & ~63ull) + 64;
            const bool planes = c->tag_pass && src.tags && plan.passes == 2 && plan.shift[0] == 32 && plan.bits[0] == 9 && plan.bits[1] == 9 &&
                                src.mode == ZK_KEYS_CANONICAL && 2 * src.K > 32 && n >= 4096 && 4 * n_pad + 2 * n + 64 <= 8 * cap;
            if (planes) {
                StreamPlanes pl{(u16*)((char*)buf_a + 4 * n_pad), plan.shift[1], plan.bits[1]};
                ZK_TRY(stream_pass0(c, src.n_bytes, src.K, src.mode, plan.shift[0], plan.bits[0], ghist, srows, first_nl, uniform, buf_a, n,
                                    c->stream_pass, &pl));
                const uint64_t blocks = 1ull << (plan.bits[0] + plan.bits[1]);
                u64* cuts;
                ZK_TRY(arena_alloc(c, sizeof(u64) * (blocks + 1), (void**)&cuts));
                ZK_TRY(stream_pass1(c, (const u32*)buf_a, pl.dig, n, ghist, 1u << plan.bits[0], plan.bits[1], (u32*)buf_b, cuts));
                src.tags->cuts = cuts; src.tags->blocks = (uint32_t)blocks; src.tags->written = true;
                *result = buf_b;
                return ZK_OK;
            }
            ZK_TRY(stream_pass0(c, src.n_bytes, src.K, src.mode, plan.shift[0], plan.bits[0], ghist, srows, first_nl, uniform, buf_a, n,
                                c->stream_pass));
#ifdef ZK_PHASES
            if (c->stream_pass >> 8) { *n_keys = 0; return ZK_OK; }          // measurement modes of the pass (diagnostic build, tools/p0_phases.py): nothing is counted
#endif
        } else
        if constexpr (C::PIPE && C::ITEMS == 16 && PipeSmem<C>::IMG_FITS && C::BLOCK <= 512) {
            // Uniform records (checked by the histogram kernel: the only newlines are one every `rec` bytes): tiles follow
            // the records, so that no key slot is spent on the windows that run into a separator (17 % of the
            // positions for 150-base reads and K = 25).  Any other stream: tiles of TILE positions.
            const uint64_t first_nl = c->h_scalars[20], nl = c->h_scalars[21], bad = c->h_scalars[22];
            if (src.mode != ZK_KEYS_BOTH && first_nl < 0x7fffffffull) {
                const uint64_t rec = first_nl + 1;
                const int64_t W = (int64_t)first_nl - src.K + 1;
                if (rec >= 16 && W >= 1 && src.n_bytes % rec == 0 && bad == 0 && nl == src.n_bytes / rec) {
                    const uint32_t cpr = (uint32_t)((W + 15) / 16);
                    const uint32_t rpt = (C::BLOCK / cpr) / 16 * 16;
                    if (rpt >= 16 && (uint64_t)rpt * rec / 16 + 3 <= (uint64_t)PipeSmem<C>::IMG_WORDS &&
                        (double)rpt * (double)W > 1.02 * (double)C::TILE * (double)W / (double)rec) {
                        a.rec = (u32)rec; a.rpt = rpt; a.cpr = cpr; a.wpr = (u32)W;
                        a.cpr_inv = (u32)(((1ull << 32) + cpr - 1) / cpr);
                    }
                }
            }
            if (src.mode != ZK_KEYS_BOTH) ZK_TRY(launch_pipe<SRC_STREAM>(c, a));
            else ZK_TRY((launch_pass<SRC_STREAM, false>(c, a)));
        } else {
            ZK_TRY((launch_pass<SRC_STREAM, false>(c, a)));
        }
        u64* in = buf_a; u64* out = buf_b;
        a.rec = 0;
        // the block dedupe comes next and a key's low 32 bits are all it needs: the last of two passes writes those only
        const bool tags = src.tags && C::PIPE && plan.passes == 2 && plan.shift[0] <= 32 && src.mode == ZK_KEYS_CANONICAL;
        for (int p = 1; p < plan.passes; p++) {
            a.kin = in; a.kout = out; a.shift = plan.shift[p]; a.bits = plan.bits[p];
            a.ghist = ghist + p * C::RADIX;
            a.tags_out = (tags && p == plan.passes - 1) ? 1 : 0;
            a.straddle = nullptr;
            // (p == 1: the pass before came from the stream and had no order to keep)
            if ((a.tags_out || p == 1) && c->tag_words >= 2 && C::PIPE) {
                // the tiles in which a bucket of the pass before begins keep their order (see pass_pipe_kernel, VAR 3): one bit each
                const u32 tile = keys_pass_tile(c, n), tiles = (u32)div_up(n, tile);
                u32* bm;
                ZK_TRY(arena_alloc(c, sizeof(u32) * (tiles / 32 + 1), (void**)&bm));
                ZK_HIP(c, hipMemsetAsync(bm, 0, sizeof(u32) * (tiles / 32 + 1), c->stream));
                hipLaunchKernelGGL(straddle_kernel, dim3((C::RADIX + 255) / 256), dim3(256), 0, c->stream, (const u64*)(ghist + (p - 1) * C::RADIX),
                                   1u << plan.bits[p - 1], tile, bm);
                ZK_HIP(c, hipGetLastError());
                a.straddle = bm;
            }
#ifdef ZK_PHASES
            if (const char* e = getenv("ZK_LOCAL_PASS")) a.dbg_local = atoi(e);
#endif
            ZK_TRY(launch_keys_pass(c, a));
#ifdef ZK_PHASES
            if (a.dbg_local) { *n_keys = 0; return ZK_OK; }          // (nothing is where it belongs: nothing may be counted from it)
#endif
            if (a.tags_out) {
                const u32 tile = keys_pass_tile(c, n), tiles = (u32)div_up(n, tile);
                const uint64_t blocks = 1ull << (plan.bits[0] + plan.bits[1]);
                u64* cuts;
                ZK_TRY(arena_alloc(c, sizeof(u64) * (blocks + 1), (void**)&cuts));
                hipLaunchKernelGGL(tag_cuts_kernel, dim3(1u << plan.bits[0]), dim3(512), 0, c->stream, (const u64*)in, (u64)n, (const u64*)ghist,
                                   (const u64*)(ghist + C::RADIX), (const u64*)c->status, (u32)C::RADIX, tile, tiles, plan.shift[1], plan.bits[1],
                                   plan.bits[0], cuts);
                ZK_HIP(c, hipGetLastError());
                src.tags->cuts = cuts; src.tags->blocks = (uint32_t)blocks; src.tags->written = true;
            }
            u64* t = in; in = out; out = t;
        }
        *result = in;
        return ZK_OK;
    }
};

// the instantiated geometries; index = zk_tune(ZK_TUNE_SORT_VARIANT / ZK_TUNE_PAIRS_VARIANT).
// Measured on MI355X (tools/sortbench.py, 10^9 50-bit keys, per pass of 16 B/key):
//   0: 8-bit digits, segmented look-back, 7 passes     1: the same with 4096-key tiles
//   2: 9-bit digits, one serial chain per digit (the first version): 2.75 TB/s x 6 passes
//   4: 9-bit digits, segmented look-back: 2.85 TB/s
//   3: 9-bit digits, persistent two-stage pipeline + scanner workgroups for the key passes: 3.07 TB/s (default)
// Tried in the pipeline and slower: 768x8 and 1024x6/8 threads x keys (2.0-2.3 TB/s: the per-wave digit
// counters cost as much as the keys), loading the next tile's keys a stage early (spills: 2.87).
typedef Cfg<512, 16, 8, 1, 4> V0;
typedef Cfg<256, 16, 8, 1, 4> V1;
typedef Cfg<512, 16, 9, 1, 4, 0> V2;
typedef Cfg<512, 16, 9, 1, 4, 32, true> V3;
typedef Cfg<512, 16, 9, 1, 4, 32> V4;
typedef Cfg<512, 16, 8, 1, 4, 32, true> V5;      // the pipeline with 8-bit digits: 256-byte runs, 7 passes
// 6: the pipeline with 16 K-key tiles, one 1024-thread workgroup per CU (150 KB LDS): a digit gets 32 keys = 256 bytes per tile
//    instead of 128 -- 3.78 vs 3.42 TB/s per pass at 2 x 10^9 keys, 3.62 vs 4.15 ms at 0.9 x 10^9.  Round 2 measured 30.1 vs 28.0 ms
//    at 6.2 x 10^9 keys and kept the 8 K tiles above 3 x 2^30 keys; round 4 measured again: whole keys, 5 x 10^9: 23.5 vs 26.3 ms
//    a pass; the tag pass of config 2 (6.2 x 10^9 keys, places from LDS adds, VAR 3): 18.05 vs 23.6-24.5 ms -- with the adds a tile's
//    ranking no longer grows with its waves' ballots, and twice the keys share a tile's barriers.  The default (3) now uses it for
//    every array pass (zk_tune ZK_TUNE_WIDE_TILES, on); pass 0 from the stream keeps the 8 K-key tiles.
typedef Cfg<1024, 16, 9, 1, 4, 32, true> V6;
// 7 (pairs only): the pipeline with a payload -- 1024 threads x 8 pairs, 8 K-pair tiles, 115 KB of LDS, one workgroup per CU
typedef Cfg<1024, 8, 9, 1, 4, 32, true> V7;
int launch_wide_pass(zk_ctx* c, const SortArgs& a) { return Sorter<V6>::launch_pipe<SRC_ARRAY>(c, a); }
#define ZK_SORT_DISPATCH(c, CALL) ZK_SORT_DISPATCH_V((c)->sort_variant, CALL)
#define ZK_SORT_DISPATCH_V(v, CALL)                 \
    switch (v) {                                    \
        case 0: return Sorter<V0>::CALL;            \
        case 1: return Sorter<V1>::CALL;            \
        case 2: return Sorter<V2>::CALL;            \
        case 4: return Sorter<V4>::CALL;            \
        case 5: return Sorter<V5>::CALL;            \
        case 6: return Sorter<V6>::CALL;            \
        default: return Sorter<V3>::CALL;           \
    }

static int sort_keys_lsd(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, u64** result) {
    ZK_SORT_DISPATCH(c, sort_keys(c, keys, alt, n, key_bits, result));
}
static int sort_pairs_lsd(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, u64** rk, u32** rv) {
    if (c->pairs_variant == 7) return Sorter<V7>::sort_pairs(c, keys, alt, vals, valt, n, key_bits, rk, rv);
    ZK_SORT_DISPATCH_V(c->pairs_variant, sort_pairs(c, keys, alt, vals, valt, n, key_bits, rk, rv));
}

// Large arrays: LSD passes over the top bits only, then every tile sorted to the end in LDS (tilesort.hip) -- unless a block of equal
// top bits turns out too long for a tile, in which case every bit is sorted by LSD passes after all (from wherever the keys are now).
int sort_keys(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, u64** result) {
    *result = keys;
    if (n == 0) return ZK_OK;
    const int top = c->tile_sort ? tile_sort_top_bits(n, key_bits, sort_rbits(c)) : 0;
    if (top) {
        u64* res = nullptr;
        bool declined = false;
        ZK_TRY(sort_keys_upper(c, keys, alt, n, key_bits, key_bits - top, &res, ZK_PROF_PASS_KEYS));
        ZK_TRY(tile_sort(c, res, nullptr, n, key_bits, top, &declined));
        *result = res;
        if (!declined) return ZK_OK;
        return sort_keys_lsd(c, res, res == keys ? alt : keys, n, key_bits, result);
    }
    return sort_keys_lsd(c, keys, alt, n, key_bits, result);
}

int sort_pairs(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, u64** rk, u32** rv) {
    *rk = keys; *rv = vals;
    if (n == 0) return ZK_OK;
    const int top = c->tile_sort ? tile_sort_top_bits(n, key_bits, sort_pairs_rbits(c)) : 0;
    if (top) {
        u64* k1 = nullptr; u32* v1 = nullptr;
        bool declined = false;
        ZK_TRY(sort_pairs_upper(c, keys, alt, vals, valt, n, key_bits, key_bits - top, &k1, &v1));
        ZK_TRY(tile_sort(c, k1, v1, n, key_bits, top, &declined));
        *rk = k1; *rv = v1;
        if (!declined) return ZK_OK;
        return sort_pairs_lsd(c, k1, k1 == keys ? alt : keys, v1, v1 == vals ? valt : vals, n, key_bits, rk, rv);
    }
    return sort_pairs_lsd(c, keys, alt, vals, valt, n, key_bits, rk, rv);
}

// keys already ordered by their low `lo_bit` bits: LSD passes over the bits above only
int sort_keys_upper(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, int lo_bit, u64** result, int prof_tag) {
    *result = keys;
    if (n == 0 || lo_bit >= key_bits) return ZK_OK;
    ZK_SORT_DISPATCH(c, sort_keys(c, keys, alt, n, key_bits, result, lo_bit, prof_tag));
}

// ... with the digit counts already taken (dedupe_finish; default geometry only)
int sort_keys_upper_counted(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, int lo_bit, u64* counted, u64** result) {
    *result = keys;
    if (n == 0 || lo_bit >= key_bits) return ZK_OK;
    return Sorter<V3>::sort_keys(c, keys, alt, n, key_bits, result, lo_bit, ZK_PROF_PASS_PACKED, counted);
}

// pairs already ordered by their low `lo_bit` bits: LSD passes over the bits above only
int sort_pairs_upper(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, int lo_bit, u64** rk, u32** rv) {
    *rk = keys; *rv = vals;
    if (n == 0) return ZK_OK;
    if (c->pairs_variant == 7) return Sorter<V7>::sort_pairs(c, keys, alt, vals, valt, n, key_bits, rk, rv, nullptr, nullptr, 0, lo_bit);
    ZK_SORT_DISPATCH_V(c->pairs_variant, sort_pairs(c, keys, alt, vals, valt, n, key_bits, rk, rv, nullptr, nullptr, 0, lo_bit));
}

// (rc(src_k[i]), src_v[i]) sorted by key; keys/alt/vals/valt are work buffers, the source arrays are only read
// lo_bit > 0: ordered by the bits [lo_bit, 2K) only (stable passes: tile_sort's input)
int sort_pairs_mirrored(zk_ctx* c, const u64* src_k, const u32* src_v, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int K,
                        u64** rk, u32** rv, int lo_bit) {
    *rk = keys; *rv = vals;
    if (n == 0) return ZK_OK;
    // (mirrored keys are all different, and their order of arrival means nothing: `unordered`)
    if (c->pairs_variant == 7) return Sorter<V7>::sort_pairs(c, keys, alt, vals, valt, n, 2 * K, rk, rv, src_k, src_v, K, lo_bit, true);
    ZK_SORT_DISPATCH_V(c->pairs_variant, sort_pairs(c, keys, alt, vals, valt, n, 2 * K, rk, rv, src_k, src_v, K, lo_bit));
}
int sort_pairs_rbits(zk_ctx* c) { return (c->pairs_variant == 0 || c->pairs_variant == 1 || c->pairs_variant == 5) ? 8 : 9; }

// digit width of the geometry used for key arrays (the truncated sort sizes its bit range with it)
// the digits sort_keys_upper / sort_keys_upper_counted will use for the bits [lo_bit, key_bits)
PassPlan sort_plan_upper(zk_ctx* c, int key_bits, int lo_bit) { return make_plan(key_bits - lo_bit, sort_rbits(c), lo_bit); }

int sort_rbits(zk_ctx* c) {
    switch (c->sort_variant) { case 0: case 1: case 5: return 8; default: return 9; }
}

// width of the first digit of sort_keys_upper(key_bits, lo_bit) -- collapse_pass must group by exactly that digit
int sort_first_bits(zk_ctx* c, int key_bits, int lo_bit) {
    if (lo_bit >= key_bits) return 0;
    return make_plan(key_bits - lo_bit, sort_rbits(c), lo_bit).bits[0];
}

// keys[0..n) ordered by their TOP b bits (of key_bits) -> the distinct keys with their counts, SORTED.  Two steps, because the
// caller can only size the result once the first is done:
//   dedupe_pass   counts the blocks (dedupe_kernel): words in `work` (at least as many words as keys), block by block.
//                 *flags: bit 0 = some table filled up (the words are not to be used), bit 1 = some counts went to the side list.
//                 max_chunks > 0: only the leading blocks (the sample; *n_in = the keys they cover).
//   dedupe_finish moves the words together and apart into out_k / out_c (r.n_out entries each).
__global__ void expand_tags_kernel(const u32* __restrict__ tags, const u64* __restrict__ cuts, u64 first_block, u64 n_blocks, int tag_bits, u64* __restrict__ out) {
    for (u64 v = first_block + blockIdx.x; v < first_block + n_blocks; v += gridDim.x) {
        const u64 lo = cuts[v], hi = cuts[v + 1], top = v << tag_bits;
        for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) out[i] = top | (u64)tags[i];
    }
}

int expand_tags(zk_ctx* c, const u32* tags, const u64* cuts, uint32_t blocks, int tag_bits, u64* keys_out, uint64_t first_block, uint64_t n_blocks) {
    if (n_blocks == 0) { first_block = 0; n_blocks = blocks; }
    const uint64_t mx = (uint64_t)c->num_cus * 16;
    hipLaunchKernelGGL(expand_tags_kernel, dim3((u32)(n_blocks < mx ? n_blocks : mx)), dim3(256), 0, c->stream, tags, cuts, (u64)first_block, (u64)n_blocks,
                       tag_bits, keys_out);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int dedupe_pass(zk_ctx* c, const u64* keys, uint64_t n, int key_bits, int b, int pack, u64* work, uint64_t cap, DedupeResult* r,
                uint64_t* n_in, uint64_t max_chunks, const u32* tags, const u64* tag_cuts) {
    *r = DedupeResult();
    if (n_in) *n_in = n;
    if (n == 0) return ZK_OK;
    if (b < 1 || b > 24 || b >= key_bits || pack < 10 || pack > 31) return fail(c, ZK_EINTERNAL, "dedupe_pass: %d block bits of %d, pack %d", b, key_bits, pack);
    if (cap < n) return fail(c, ZK_ENOSPC, "dedupe_pass: work buffer of %llu words for %llu keys", (unsigned long long)cap, (unsigned long long)n);
    DedupeArgs a = {};
    uint64_t chunks = 1ull << b;          // one block per value of the top bits
    if (max_chunks && chunks > max_chunks) chunks = max_chunks;
    u64 *cuts, *nwords, *incl, *big;
    const u32 big_cap = 1u << 16;
    if (tags) {
        if (!tag_cuts || key_bits - b > 32) return fail(c, ZK_EINTERNAL, "dedupe_pass: tags of %d bits", key_bits - b);
        cuts = const_cast<u64*>(tag_cuts);
    } else ZK_TRY(arena_alloc(c, sizeof(u64) * (chunks + 1), (void**)&cuts));
    ZK_TRY(arena_alloc(c, sizeof(u64) * chunks, (void**)&nwords));
    ZK_TRY(arena_alloc(c, sizeof(u64) * chunks, (void**)&incl));
    ZK_TRY(arena_alloc(c, sizeof(u64) * 2 * big_cap, (void**)&big));
    a.tag_bits = key_bits - b;
    if (!tags) hipLaunchKernelGGL(dedupe_cuts_kernel, dim3((u32)div_up(chunks + 1, 256)), dim3(256), 0, c->stream, keys, (u64)n, a.tag_bits, (u32)chunks, cuts);
    a.kin = keys; a.tin = tags; a.n = n; a.cuts = cuts; a.out = work; a.nwords = nwords; a.pack = pack;
    a.chunks = (u32)chunks;
    a.flags = (u32*)(c->d_scalars + 27);
    a.counter = (u32*)(c->d_scalars + 29);
    a.n_big = (u32*)(c->d_scalars + 30);
    a.big = big; a.big_cap = big_cap;
    const u32 bad_cap = 64;
    ZK_TRY(arena_alloc(c, sizeof(u32) * bad_cap, (void**)&a.bad));
    a.bad_cap = bad_cap;
    a.n_bad = (u32*)(c->d_scalars + 31);
    a.dbg = c->dbg ? c->dbg + 8192 : nullptr;
    // the mirror sort can group by 6 more bits if the blocks say how their entries split on them: 64 counts per block, when the
    // workspace has the room (and the finer grouping's tables after it: dedupe_finish)
    // ... leaving what the sorts and the union after it need (their tables are a few bytes per thousand keys)
    if (!max_chunks && a.tag_bits >= 14 && c->arena_size - c->arena_off > 64ull * chunks * (4 + 8 + 8) + (32ull << 20) + n / 16)
        ZK_TRY(arena_alloc(c, sizeof(u32) * 64 * chunks, (void**)&a.sub));
    ZK_HIP(c, hipMemsetAsync(c->d_scalars + 27, 0, 5 * sizeof(u64), c->stream));
    // algorithmic bytes: every key read once (a 32-bit tag, or the whole key), one word written per distinct key (added below, once
    // the launch has said how many)
    prof_begin(c, ZK_PROF_RLE, (tags ? 4 : 8) * n);
    auto launch_one_per_cu = [&](const DedupeArgs& d) {
        const u32 grid = d.chunks < (u32)c->num_cus ? d.chunks : (u32)c->num_cus;
        if (tags) hipLaunchKernelGGL((dedupe_kernel<true, true>), dim3(grid), dim3(1024), 0, c->stream, d);
        else if (d.tag_bits <= 32) hipLaunchKernelGGL((dedupe_kernel<true, false>), dim3(grid), dim3(1024), 0, c->stream, d);
        else hipLaunchKernelGGL((dedupe_kernel<false, false>), dim3(grid), dim3(1024), 0, c->stream, d);
    };
    const bool two_per_cu = a.tag_bits <= 32 && c->dedupe_variant >= 0;
    if (two_per_cu) {
        ZK_TRY(arena_alloc(c, sizeof(u32) * chunks, (void**)&a.retry));
        a.n_retry = (u32*)(c->d_scalars + 28);
        a.limit = (u32)c->dedupe_limit;
        ZK_TRY(launch_dedupe2(c, a, tags != nullptr, c->dedupe_variant));
    } else launch_one_per_cu(a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 27, c->d_scalars + 27, 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 32, cuts + chunks, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_TRY(check_device_error(c));
    if (two_per_cu && (uint32_t)c->h_scalars[28]) {
        // the blocks dedupe2_kernel declined (65 536 keys or more; a table that filled up): dedupe_kernel's table is larger and its
        // counts are 32 bits wide -- what it declines too goes on the list the host counts by sorting
        DedupeArgs d = a;
        d.list = a.retry; d.chunks = (uint32_t)c->h_scalars[28]; d.retry = nullptr; d.n_retry = nullptr;
        ZK_HIP(c, hipMemsetAsync(a.counter, 0, sizeof(u32), c->stream));
        launch_one_per_cu(d);
        ZK_HIP(c, hipGetLastError());
        ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 27, c->d_scalars + 27, 5 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        ZK_TRY(check_device_error(c));
    }
    r->flags = (uint32_t)c->h_scalars[27];
    r->n_big = (uint32_t)c->h_scalars[30];
    if (r->n_big > big_cap) r->flags |= 1;          // more counts beyond the field than the side list holds: the long way
    const uint32_t n_bad = (uint32_t)c->h_scalars[31];
    if (n_bad && n_bad <= bad_cap && !(r->flags & 1) && !max_chunks) {
        // The few blocks whose table filled up (a stretch of the key space with more distinct k-mers than a table holds) are
        // counted the plain way, one by one: their keys sorted by the bits below the block bits (the block's own place in `work`
        // is the second buffer), run lengths into words at that place, the block's word count patched in.
        uint32_t list[64];
        ZK_HIP(c, hipMemcpy(list, a.bad, sizeof(u32) * n_bad, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_bad && !(r->flags & 1); i++) {
            u64 lohi[2];
            ZK_HIP(c, hipMemcpy(lohi, cuts + list[i], 2 * sizeof(u64), hipMemcpyDeviceToHost));
            const uint64_t m = lohi[1] - lohi[0];
            u64* res = nullptr;
            u64* bk = const_cast<u64*>(keys) + lohi[0];          // the block's keys, sorted in place
            if (tags) {
                // only tags were written: the block's keys are made again, beside the lists (a block is a few thousand keys)
                ZK_TRY(arena_alloc(c, 8 * m, (void**)&bk));
                ZK_TRY(expand_tags(c, tags, cuts, (uint32_t)chunks, a.tag_bits, bk - lohi[0], list[i], 1));
            }
            ZK_TRY(sort_keys(c, bk, work + lohi[0], m, a.tag_bits, &res));
            if (res != bk) ZK_HIP(c, hipMemcpyAsync(bk, res, 8 * m, hipMemcpyDeviceToDevice, c->stream));
            uint64_t u = 0;
            bool ovf = false;
            ZK_TRY(rle(c, bk, m, work + lohi[0], nullptr, m, &u, pack, &ovf));
            if (ovf) { r->flags |= 1; break; }          // (a count beyond the field in such a block: the long way after all)
            ZK_HIP(c, hipMemcpy(nwords + list[i], &u, sizeof(u64), hipMemcpyHostToDevice));
        }
        a.sub = nullptr;          // the runs of those blocks were not counted: the mirror sort groups by the block bits only
    } else if (n_bad > bad_cap) r->flags |= 1;
    ZK_HIP(c, hipMemcpyAsync(incl, nwords, sizeof(u64) * chunks, hipMemcpyDeviceToDevice, c->stream));
    ZK_TRY(scan64_inclusive(c, incl, chunks));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, incl + chunks - 1, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    r->n_out = c->h_scalars[9];
    prof_add_bytes(c, ZK_PROF_RLE, 8 * r->n_out);          // one word written per distinct key
    r->cuts = cuts; r->nwords = nwords; r->incl = incl; r->big = big; r->chunks = (uint32_t)chunks; r->pack = pack; r->work = work; r->sub = a.sub;
    if (n_in) *n_in = c->h_scalars[32];          // keys covered by the blocks that were counted
    return ZK_OK;
}

// out_m (or null; K odd or even, 2 * gbases block bits = all 4^gbases blocks counted): the mirrored words, grouped by their low
// 2 * gbases bits (dedupe_unpack_kernel) -- ready for the passes over the bits above
int dedupe_finish(zk_ctx* c, const DedupeResult& r, u64* out_k, u32* out_c, u64* out_m, int K, int gbases, u64** mirror_hist,
                  int* mirror_group_bits, bool packed_out) {
    if (mirror_hist) *mirror_hist = nullptr;
    int gbits = 2 * gbases;
    if (mirror_group_bits) *mirror_group_bits = gbits;
    if (r.n_out == 0) return ZK_OK;
    u64 *minc = nullptr, *place24 = nullptr;
    if (out_m && (1ull << (2 * gbases)) != r.chunks) return fail(c, ZK_EINTERNAL, "dedupe_finish: %u blocks are not 4^%d", r.chunks, gbases);
    if (out_m && r.sub && mirror_group_bits && 2 * K - gbits - 6 >= 8) {
        // 6 more group bits: one pass less for the mirror sort (26 bits above the groups instead of 32 at K = 25)
        const uint64_t runs = 64ull * r.chunks;
        u64* minc24;
        ZK_TRY(arena_alloc(c, sizeof(u64) * runs, (void**)&minc24));
        ZK_TRY(arena_alloc(c, sizeof(u64) * runs, (void**)&place24));
        hipLaunchKernelGGL(dedupe_mirror_sizes24_kernel, dim3((u32)div_up(runs, 256)), dim3(256), 0, c->stream, r.sub, r.chunks, gbases, minc24);
        ZK_TRY(scan64_inclusive(c, minc24, runs));
        hipLaunchKernelGGL(dedupe_mirror_place24_kernel, dim3((u32)div_up(runs, 256)), dim3(256), 0, c->stream, r.sub, minc24, r.chunks, gbases, place24);
        ZK_HIP(c, hipGetLastError());
        gbits += 6;
        *mirror_group_bits = gbits;
    }
    MirrorHist mh = {};
    if (out_m && mirror_hist && c->sort_variant == 3) {
        // the digit counts of the passes that will sort the mirrored words above their group bits (sort_keys_upper_counted)
        const PassPlan plan = make_plan(2 * K - gbits, V3::RBITS, gbits + r.pack);
        if (plan.passes <= 4) {
            ZK_TRY(arena_alloc(c, sizeof(u64) * MAX_PASSES * V3::RADIX, (void**)&mh.raw));
            ZK_HIP(c, hipMemsetAsync(mh.raw, 0, sizeof(u64) * MAX_PASSES * V3::RADIX, c->stream));
            mh.passes = plan.passes;
            for (int p = 0; p < plan.passes; p++) { mh.shift[p] = plan.shift[p]; mh.bits[p] = plan.bits[p]; }
            *mirror_hist = mh.raw;
        }
    }
    if (out_m && !place24) {
        ZK_TRY(arena_alloc(c, sizeof(u64) * r.chunks, (void**)&minc));
        hipLaunchKernelGGL(dedupe_mirror_sizes_kernel, dim3((r.chunks + 255) / 256), dim3(256), 0, c->stream, r.nwords, r.chunks, gbases, minc);
        ZK_TRY(scan64_inclusive(c, minc, r.chunks));
    }
    if (packed_out && r.n_big) return fail(c, ZK_EINTERNAL, "dedupe_finish: packed words with %u counts beyond the field", r.n_big);
    prof_begin(c, ZK_PROF_SELECT, ((out_m ? 28 : 20) - (packed_out ? 4 : 0)) * r.n_out);
    hipLaunchKernelGGL(dedupe_unpack_kernel, dim3((u32)c->num_cus * 8), dim3(256), 0, c->stream, r.work, r.cuts, r.incl, r.nwords, r.chunks, r.pack, out_k, out_c,
                       out_m, minc, K, gbases, mh, place24, packed_out ? 1 : 0);
    if (r.n_big) hipLaunchKernelGGL(dedupe_big_kernel, dim3((r.n_big + 255) / 256), dim3(256), 0, c->stream, r.big, r.n_big, out_k, (u64)r.n_out, out_c, c->d_err);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

// keys[0..n) ordered by their low `shift` bits -> out[0..*n_out): one word (key << pack | run length) per run of equal keys
// inside a tile grouped by the digit [shift, shift + bits); max_tiles > 0: only the leading tiles, at most that many (the
// sample).  2^pack must exceed 512 (run pieces are cut at 512 when 2^pack <= 8192).
int collapse_pass(zk_ctx* c, const u64* keys, uint64_t n, int shift, int bits, int pack, u64* out, uint64_t cap, uint64_t* n_out,
                  uint64_t max_tiles) {
    *n_out = 0;
    if (n == 0) return ZK_OK;
    if (bits < 1 || bits > 9 || pack < 10 || pack > 31) return fail(c, ZK_EINTERNAL, "collapse_pass: bits %d, pack %d", bits, pack);
    CollapseArgs a = {};
    constexpr uint64_t TILE = CollapseSmem<9>::TILE;
    uint64_t tiles = div_up(n, TILE);
    if (max_tiles && tiles > max_tiles) { tiles = max_tiles; n = tiles * TILE; }
    a.kin = keys; a.n = n; a.out = out; a.cap = cap; a.shift = shift; a.bits = bits; a.pack = pack;
    a.split = (1u << pack) <= (u32)TILE;
    a.tiles = (u32)tiles;
    ZK_TRY(lookback_begin(c, tiles, (u32)tiles, &a.epoch, &a.ticket_base));
    a.status = c->status; a.ticket = c->d_ticket; a.err = c->d_err; a.d_total = c->d_scalars + 9;
    prof_begin(c, ZK_PROF_RLE, 8 * n);
    if (bits <= 8) hipLaunchKernelGGL(collapse_kernel<8>, dim3((u32)tiles), dim3(512), 0, c->stream, a);
    else hipLaunchKernelGGL(collapse_kernel<9>, dim3((u32)tiles), dim3(512), 0, c->stream, a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_TRY(check_device_error(c));
    *n_out = c->h_scalars[9];
    return ZK_OK;
}

int sort_stream(zk_ctx* c, const StreamSrc& src, u64* buf_a, u64* buf_b, uint64_t cap, uint64_t* n_keys,
                uint64_t acgt[4], u64** result) {
    *result = buf_a;
    *n_keys = 0;
    if (acgt) acgt[0] = acgt[1] = acgt[2] = acgt[3] = 0;
    if (src.n_bytes == 0) return ZK_OK;
    if ((uintptr_t)src.stream & 15) return fail(c, ZK_EINVAL, "base stream must be 16-byte aligned");
    ZK_SORT_DISPATCH(c, sort_stream(c, src, buf_a, buf_b, cap, n_keys, acgt, result));
}

}  // namespace zk
