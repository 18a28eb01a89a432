// stream_pass.hip -- the histogram and the first pass of the k-mer sort, as a count / scan / scatter pass over STATIC ranges.
//
// Replaces the first pass of misc.radix_sort (zotmer/library/misc.py:400-424) over the k-mers that basics.kmersList
// (zotmer/library/basics.py:303-347) makes of the reads (commands/kmerize.py:490-493, 412-417).
//
// Why not the look-back pipeline of radix_sort.hip for this pass: what bounds a scatter pass on MI355X is who writes which
// 128-byte line (tools/scatter_bench2.hip, profiles/r03/scatter_patterns.json).  6.2 G keys into 512 streams take 10-12 ms when
// every 64- or 128-byte unit is written whole by ONE wave instruction, and 20-25 ms when the units are cut by run borders -- which
// is what a tile-by-tile pass does: tile t's 16 keys of a digit land wherever the counts of the tiles before it say.  The first
// pass has no order to keep (its input is the read stream), so here
//   * the stream is cut into one contiguous RANGE per workgroup (one workgroup per CU);
//   * the histogram kernel, which reads the stream anyway, also counts the digits of this pass PER RANGE (rows);
//   * a scan gives every (range, digit) its own contiguous piece of the output;
//   * the pass kernel walks its range tile by tile, parks each tile grouped by digit in LDS, and writes only WHOLE units of G
//     keys (G * 8 bytes, aligned in the output); what is left of a digit (fewer than G keys) waits in LDS for the next tile.
// No look-back, no scanner workgroups, no status words, no atomics outside LDS.
//
// The same file holds the leaner histogram kernel: bytes -> 2-bit codes four at a time (SWAR), the canonical strand decided on
// the words that hold the digits, acgt from population counts instead of per-window adds.
#include "internal.hpp"
#include "encode_tile.hpp"

#include <stdlib.h>
#include <type_traits>

namespace zk {

constexpr int P0_BLOCK = 512, P0_NW = 8, P0_TILE = P0_BLOCK * P0_NW, P0_IMG_WORDS = 320;          // the pass: 2 workgroups per CU
constexpr int SH_BLOCK = 512, SH_NW = 16, SH_TILE = SH_BLOCK * SH_NW;          // the histogram kernel's tiles: positions

// How the stream is cut.  Decided from the position of the stream's first newline alone (the histogram kernel runs before
// anybody knows whether the records are uniform), by the same function on the device and on the host.
struct StreamTiling {
    u32 rec;              // bytes per record (bases + separator) if the stream might consist of uniform records, else 0
    u32 rpt, cpr, wpr;    // records per tile, threads per record, windows per record
    u32 cpr_inv;          // ceil(2^32 / cpr)
    u32 tile_bytes;       // stream bytes per tile of the pass (rpt * rec, or P0_TILE positions)
    u32 ranges;
    u64 range_bytes;      // stream bytes per range: a whole number of tiles, a multiple of 16
};

__host__ __device__ inline StreamTiling make_tiling(u64 n_bytes, u64 first_nl, int K, u32 ranges) {
    StreamTiling t = {};
    t.ranges = ranges ? ranges : 1;
    t.tile_bytes = P0_TILE;
    if (first_nl < 0x7fffffffull) {
        const u64 rec = first_nl + 1;
        const long long W = (long long)first_nl - K + 1;
        if (rec >= 16 && W >= 1 && n_bytes % rec == 0) {
            const u32 cpr = (u32)((W + P0_NW - 1) / P0_NW);
            u32 rpt = cpr <= (u32)P0_BLOCK ? ((u32)P0_BLOCK / cpr) / 16 * 16 : 0;
            const u32 fit = (u32)((16ull * (P0_IMG_WORDS - 3)) / rec) / 16 * 16;
            if (rpt > fit) rpt = fit;
            // tiles that follow the records spend no key slot on the windows that run into a separator: worth it when such a
            // tile covers more of the stream than a tile of positions
            if (rpt >= 16 && (u64)rpt * rec * 50 > (u64)P0_TILE * 51) {
                t.rec = (u32)rec; t.rpt = rpt; t.cpr = cpr; t.wpr = (u32)W;
                t.cpr_inv = (u32)(((1ull << 32) + cpr - 1) / cpr);
                t.tile_bytes = rpt * (u32)rec;
            }
        }
    }
    const u64 tiles = (n_bytes + t.tile_bytes - 1) / t.tile_bytes;
    const u64 tpr = (tiles + t.ranges - 1) / t.ranges;
    t.range_bytes = (tpr ? tpr : 1) * (u64)t.tile_bytes;
    return t;
}

// ---------------------------------------------------------------------------------------
// bytes -> 2-bit codes and validity, four bytes per operation
// ---------------------------------------------------------------------------------------
// 16 stream bytes -> codes (2 bits per base, first base in the top bits) and validity (16 bits, first base in bit 15); the same
// function of the bytes as encode_words16 (A a C c G g T t U u are bases, zotmer/library/basics.py:42-46).
__device__ __forceinline__ void encode_swar16(const uint4 q, u32& codes, u32& vmask) {
    const u32 w[4] = {q.x, q.y, q.z, q.w};
    u32 cc = 0, vv = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // codes: bits 2..1 of the byte, A0 C1 G3 T2 -> A0 C1 G2 T3
        const u32 t = (w[i] >> 1) & 0x03030303u;
        const u32 code = t ^ ((t >> 1) & 0x01010101u);
        const u32 g = (code | (code >> 6) | (code >> 12) | (code >> 18)) & 0xffu;          // byte j's code at bits 2j
        cc |= g << (8 * i);
        // validity: x = byte | 0x20 is one of 0x61 0x63 0x67 0x74 0x75:
        //   ~b7 & b6 & ~b3 & (b4 ? (b2 & ~b1) : (b0 & ~(b2 & ~b1)))
        const u32 x = w[i] | 0x20202020u;
        const u32 t1 = (x >> 2) & ~(x >> 1);
        const u32 b4 = x >> 4;
        const u32 inner = (b4 & t1) | (~b4 & x & ~t1);
        const u32 ok = ~(x >> 7) & (x >> 6) & ~(x >> 3) & inner & 0x01010101u;
        const u32 n = (ok | (ok >> 7) | (ok >> 14) | (ok >> 21)) & 0xfu;                   // byte j's bit at bit j
        vv |= n << (4 * i);
    }
    // first base low -> first base high
    const u32 y = __brev(cc);
    codes = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    vmask = __brev(vv) >> 16;
}

// (a : b) >> t, 0 <= t < 128 (per lane): the low three 32-bit words
__device__ __forceinline__ void shr128(u64 a, u64 b, u32 t, u32& w0, u32& w1, u32& w2) {
    const u32 s = t & 63u;
    const u64 lo_small = s ? ((b >> s) | (a << (64 - s))) : b;          // t < 64
    const u64 hi_small = a >> s;
    const bool big = t >= 64;
    const u64 lo = big ? hi_small : lo_small;
    const u64 hi = big ? 0ull : hi_small;
    w0 = (u32)lo; w1 = (u32)(lo >> 32); w2 = (u32)hi;
}

// The NW windows that start at string position s .. s + NW - 1 of the 64 bases in (a : b), as three-word strings from which every
// window is two funnel shifts by a CONSTANT: forward  x_i = (S' >> 2 (NW - 1 - i)) & mask,  reverse complement  xb_i = (R' >> 2 i) & mask.
template <int NW>
struct WindowWords {
    u32 s0, s1, s2;       // S' = S >> (128 - 2 s - 2 (NW - 1) - 2 K)
    u32 r0, r1, r2;       // R' = revcomp(S) >> 2 s
    __device__ __forceinline__ void init(u64 a, u64 b, u32 s, int K) {
        shr128(a, b, 128u - 2u * s - 2u * (NW - 1) - 2u * (u32)K, s0, s1, s2);
        const u64 rhi = rev_pairs(~b), rlo = rev_pairs(~a);
        shr128(rhi, rlo, 2u * s, r0, r1, r2);
    }
    __device__ __forceinline__ u32 xhi(int i) const { return __builtin_amdgcn_alignbit(s2, s1, 2 * (NW - 1 - i)); }
    __device__ __forceinline__ u32 xlo(int i) const { return __builtin_amdgcn_alignbit(s1, s0, 2 * (NW - 1 - i)); }
    __device__ __forceinline__ u32 bhi(int i) const { return __builtin_amdgcn_alignbit(r2, r1, 2 * i); }
    __device__ __forceinline__ u32 blo(int i) const { return __builtin_amdgcn_alignbit(r1, r0, 2 * i); }
};

// bit 63 - q of the result: the K bases from string position q on are all valid
__device__ __forceinline__ u64 runs_of_k(u64 v, int K) {
    int have = 1;
    while (2 * have <= K) { v &= v << have; have *= 2; }
    v &= v << (K - have);
    return v;
}

// bit k of a 16-bit value -> bit 2 k
__device__ __forceinline__ u32 spread16(u32 x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// ---------------------------------------------------------------------------------------
// the histogram kernel
// ---------------------------------------------------------------------------------------
struct SHistArgs {
    const u8* stream;
    u64 n_bytes;
    int K, mode;
    PassPlan plan;
    u32 binbase[MAX_PASSES];      // where pass p's bins start in LDS
    u32 nbins;
    u64* ghist;                   // [passes][gstride] += digit counts over the whole stream
    u32 gstride;
    u32* rows;                    // [ranges][1 << plan.bits[0]] += pass 0's digit counts per range
    u64* acgt;                    // [4] or null
    u64* rec_info;                // [0] = position of the stream's first newline (read); [1] += newline bytes, [2] += bad chunks
    u64* sample;                  // or null: keys with (key >> sample_shift) == sample_value are also appended here ...
    u32* sample_n;                // ... += 1 each (appended while below sample_cap)
    u32 sample_cap;
    int sample_shift;
    u64 sample_value;
    u32 revbase;                  // ZK_KEYS_BOTH: where the bins of pass 0's digit of the reverse strand start in LDS (rows [ranges + w])
    u32 ranges, split;            // grid = ranges * split: workgroup (w, s) takes every split-th tile of range w
    u32* gcodes;                  // [ceil(n_bytes / 16)] the stream's 2-bit image, 16 bases per word (first base on top) ...
    u16* gvalid;                  // ... and which of them are bases: the pass reads these instead of encoding the bytes again
    int dbg_atomics;              // diagnostic build (-DZK_PHASES) only: LDS adds per window (2 = as in the product)
};

// HI: every digit (and the sample test) looks only at key bits >= 32: the strands are compared on the high words alone --
// hi(min(x, xb)) == min(hi x, hi xb) -- and the low words are never built.
// TWO: the plan has exactly two passes (the usual one: two passes over the top bits, then the block dedupe), their digits in
// scalar registers; otherwise the passes are walked through a table in LDS.
// CANON_T: 1 = the mode is known to be ZK_KEYS_CANONICAL when the kernel is compiled (the usual launch), -1 = read from the arguments.
template <bool HI, bool TWO, int CANON_T = -1>
__global__ __launch_bounds__(SH_BLOCK, 4) void stream_hist_kernel(SHistArgs h) {
    extern __shared__ u32 bins[];
    __shared__ TileImage<SH_TILE> img;
    __shared__ u32 pshift[MAX_PASSES], pmask[MAX_PASSES], pbase[MAX_PASSES];
    const int tid = threadIdx.x;
    for (u32 i = tid; i < h.nbins; i += SH_BLOCK) bins[i] = 0;
    if (tid < MAX_PASSES) {
        pshift[tid] = (u32)h.plan.shift[tid] - (HI ? 32u : 0u);
        pmask[tid] = (1u << h.plan.bits[tid]) - 1u;
        pbase[tid] = h.binbase[tid];
    }
    const u64 first_nl = h.rec_info[0];
    const StreamTiling tl = make_tiling(h.n_bytes, first_nl, h.K, h.ranges);
    const u32 w = blockIdx.x / h.split, sp = blockIdx.x % h.split;
    const u64 B = (u64)w * tl.range_bytes;
    const u64 E = (B + tl.range_bytes < h.n_bytes) ? B + tl.range_bytes : h.n_bytes;
    const u32 ntile = B < E ? (u32)((E - B + SH_TILE - 1) / SH_TILE) : 0u;
    const int K = h.K;
    const int np = h.plan.passes;
    const u64 mask = ~0ull >> (64 - 2 * K);
    const u32 mlo = (u32)mask, mhi = (u32)(mask >> 32);
    const bool canon = CANON_T == 1 ? true : h.mode == ZK_KEYS_CANONICAL;
    // ZK_KEYS_BOTH: x and rc x are two keys.  The pass takes the two strands of a range as two ranges of their own (w: x, ranges + w:
    // rc x), so pass 0's digit is counted apart for them; only the table-driven form (not TWO) knows this mode.
    const bool both = CANON_T != 1 && h.mode == ZK_KEYS_BOTH;
    // TWO: the two digits
    const u32 sh0 = (u32)h.plan.shift[0] - (HI ? 32u : 0u), sh1 = (u32)h.plan.shift[1] - (HI ? 32u : 0u);
    const u32 dm0 = (1u << h.plan.bits[0]) - 1u, dm1 = (1u << h.plan.bits[1]) - 1u;
    const u32 base1 = h.binbase[1];
    const u32 nb0 = (u32)h.plan.bits[0], nb1 = (u32)h.plan.bits[1];
    const u32 ssh = (u32)h.sample_shift - (HI ? 32u : 0u);
    const typename std::conditional<HI, u32, u64>::type sval = (typename std::conditional<HI, u32, u64>::type)h.sample_value;
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0, nl_count = 0, bad_count = 0;
    // are the records uniform?  every chunk counts its newline bytes and whether they are exactly the expected separator
    u32 rec = 0, m = 0, step_m = 0;
    if (first_nl < 0x7fffffffull) {
        rec = (u32)first_nl + 1u;
        m = (u32)((B + (u64)sp * SH_TILE + 16ull * tid) % rec);
        step_m = (u32)(((u64)h.split * SH_TILE) % rec);
    }
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    auto issue = [&](u32 j) {
        const u64 P = B + (u64)j * SH_TILE;
        q0 = load_chunk16(h.stream, h.n_bytes, P + 16ull * tid);
        if (tid < 3) q1 = load_chunk16(h.stream, h.n_bytes, P + 16ull * (tid + SH_BLOCK));
    };
    if (sp < ntile) issue(sp);
    __syncthreads();
    for (u32 j = sp; j < ntile; j += h.split) {
        const u64 off = B + (u64)j * SH_TILE + 16ull * tid;
        const bool mine = off < E;          // chunks past the range's end belong to the next range
        {
            u32 cc, vv;
            encode_swar16(q0, cc, vv);
            if (rec) {
                if (mine) {
                    // Is this chunk's only newline the separator it should have?  If every byte but that one is a base (the usual
                    // chunk), only the separator's byte needs a look; a chunk with any other non-base (an N, the stream's end) is
                    // checked byte by byte.
                    const int sep = (int)(rec - 1u) - (int)m;
                    const bool expect = sep >= 0 && sep < 16 && off + (u64)sep < h.n_bytes;
                    const u32 sepbit = expect ? (0x8000u >> (sep & 15)) : 0u;
                    if ((vv | sepbit) == 0xffffu) {
                        const u32 wsel = (sep & 12) == 0 ? q0.x : (sep & 12) == 4 ? q0.y : (sep & 12) == 8 ? q0.z : q0.w;
                        const bool isnl = ((wsel >> (8 * (sep & 3))) & 0xffu) == 0x0au;
                        nl_count += (expect && isnl) ? 1u : 0u;
                        bad_count += (expect && !isnl) ? 1u : 0u;
                    } else {
                        const u32 w4[4] = {q0.x, q0.y, q0.z, q0.w};
                        bool bad = false;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const u32 x = w4[q] ^ 0x0a0a0a0au;
                            const u32 t = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
                            nl_count += (u32)__popc(t);
                            const u32 want = (expect && (sep >> 2) == q) ? (0x80u << (8 * (sep & 3))) : 0u;
                            bad |= (t != want);
                        }
                        bad_count += bad ? 1u : 0u;
                    }
                }
                m += step_m;
                if (m >= rec) m -= rec;
            }
            img.codes[tid] = cc; img.valid[tid] = vv;
            if (mine && h.gcodes) { h.gcodes[off >> 4] = cc; h.gvalid[off >> 4] = (u16)vv; }
            if (tid < 3) {
                encode_swar16(q1, cc, vv);
                img.codes[tid + SH_BLOCK] = cc; img.valid[tid + SH_BLOCK] = vv;
            }
        }
        __syncthreads();
        if (j + h.split < ntile) issue(j + h.split);
        const u64 a = ((u64)img.codes[tid] << 32) | img.codes[tid + 1];
        const u64 b = ((u64)img.codes[tid + 2] << 32) | img.codes[tid + 3];
        u64 v = ((u64)img.valid[tid] << 48) | ((u64)img.valid[tid + 1] << 32) | ((u64)img.valid[tid + 2] << 16) | (u64)img.valid[tid + 3];
        v = runs_of_k(v, K);
        const u32 V16 = mine ? (u32)(v >> 48) : 0u;          // bit 15 - i: window i is a k-mer of this range
        WindowWords<SH_NW> ww;
        ww.init(a, b, 0u, K);
        if (h.acgt) {
            // acgt[x & 3] and acgt[xb & 3] over the valid windows: x & 3 is the window's last base, xb & 3 the complement of its first
            const u32 Vs = spread16(V16);
            const u32 ff = (u32)(a >> 32);                                               // bases 0 .. 15
            const u32 fl = (u32)((K > 1 ? ((a << (2 * K - 2)) | (b >> (66 - 2 * K))) : a) >> 32);      // bases K - 1 .. K + 14
            const u32 lf = ff & Vs, hf = (ff >> 1) & Vs, ll = fl & Vs, hl = (fl >> 1) & Vs;
            a0 += (u32)__popc(Vs & ~hl & ~ll) + (u32)__popc(hf & lf);
            a1 += (u32)__popc(~hl & ll) + (u32)__popc(hf & ~lf);
            a2 += (u32)__popc(hl & ~ll) + (u32)__popc(~hf & lf);
            a3 += (u32)__popc(hl & ll) + (u32)__popc(Vs & ~hf & ~lf);
        }
        u32 hits = 0;          // windows whose key belongs to the set-aside blocks
#pragma unroll
        for (int i = 0; i < SH_NW; i++) {
            const u32 inc = (V16 >> (15 - i)) & 1u;
            // kd: the word the digits are taken from (HI: the key's high word)
            typename std::conditional<HI, u32, u64>::type kd;
            if constexpr (HI) {
                const u32 xh = ww.xhi(i) & mhi, bh = ww.bhi(i) & mhi;
                kd = canon ? (xh < bh ? xh : bh) : xh;
            } else {
                const u64 x = ((u64)(ww.xhi(i) & mhi) << 32) | (ww.xlo(i) & mlo), xb = ((u64)(ww.bhi(i) & mhi) << 32) | (ww.blo(i) & mlo);
                kd = canon ? (x < xb ? x : xb) : x;
            }
            if constexpr (TWO && HI) {
#ifdef ZK_PHASES          // measurement (results are wrong): ZK_HIST_ATOMICS = 0 / 1 of the two adds per window -- what bounds the kernel, the
                          // atomic unit or the window arithmetic? (tools/p0_phases.py, profiles/r04/hist_atomics.json)
                if (h.dbg_atomics >= 1) atomicAdd(&bins[__builtin_amdgcn_ubfe(kd, sh0, nb0)], inc);
                if (h.dbg_atomics >= 2) atomicAdd(&bins[base1 + __builtin_amdgcn_ubfe(kd, sh1, nb1)], inc);
                if (h.dbg_atomics == 0) a0 += kd & inc;          // (the key stays live)
#else
                atomicAdd(&bins[__builtin_amdgcn_ubfe(kd, sh0, nb0)], inc);
                atomicAdd(&bins[base1 + __builtin_amdgcn_ubfe(kd, sh1, nb1)], inc);
#endif
            } else if constexpr (TWO) {
                atomicAdd(&bins[(u32)(kd >> sh0) & dm0], inc);
                atomicAdd(&bins[base1 + ((u32)(kd >> sh1) & dm1)], inc);
            } else {
                for (int p = 0; p < np; p++) atomicAdd(&bins[pbase[p] + ((u32)(kd >> pshift[p]) & pmask[p])], inc);
                if (both) {
                    typename std::conditional<HI, u32, u64>::type kr;
                    if constexpr (HI) kr = ww.bhi(i) & mhi;
                    else kr = ((u64)(ww.bhi(i) & mhi) << 32) | (ww.blo(i) & mlo);
                    atomicAdd(&bins[h.revbase + ((u32)(kr >> pshift[0]) & pmask[0])], inc);
                    for (int p = 1; p < np; p++) atomicAdd(&bins[pbase[p] + ((u32)(kr >> pshift[p]) & pmask[p])], inc);
                }
            }
            if constexpr (HI) {
                if (h.sample) hits |= ((kd >> ssh) == sval ? 1u : 0u) << (15 - i);          // looked at once per tile, below
            } else if (h.sample && inc && (kd >> ssh) == sval) {
                const u32 at = atomicAdd(h.sample_n, 1u);
                if (at < h.sample_cap) h.sample[at] = kd;
            }
        }
        hits &= V16;
        if (HI && hits) {          // rare (four blocks of 2^18): the set-aside keys are made in full and appended
#pragma unroll
            for (int i = 0; i < SH_NW; i++) {
                if ((hits >> (15 - i)) & 1u) {
                    const u64 x = (((u64)ww.xhi(i) << 32) | ww.xlo(i)) & mask, xb = (((u64)ww.bhi(i) << 32) | ww.blo(i)) & mask;
                    const u64 kk = canon ? (x < xb ? x : xb) : x;
                    const u32 at = atomicAdd(h.sample_n, 1u);
                    if (at < h.sample_cap) h.sample[at] = kk;
                }
            }
        }
        __syncthreads();          // the image is restaged by the next iteration
    }
    __syncthreads();
    const u32 r0 = 1u << h.plan.bits[0];
    for (u32 i = tid; i < h.nbins; i += SH_BLOCK) {
        const u32 c = bins[i];
        if (!c) continue;
        if (i < r0) atomicAdd(&h.rows[(u64)w * r0 + i], c);
        if (both && i >= h.revbase) {          // pass 0's digit, reverse strand
            atomicAdd(&h.rows[(u64)(h.ranges + w) * r0 + (i - h.revbase)], c);
            atomicAdd(&h.ghist[i - h.revbase], (u64)c);
            continue;
        }
        int p = 0;          // the pass bin i belongs to
        for (int q = 1; q < np; q++) if (i >= pbase[q]) p = q;
        atomicAdd(&h.ghist[(u64)p * h.gstride + (i - pbase[p])], (u64)c);
    }
    if (rec) {
        nl_count = wave_sum_u32(nl_count); bad_count = wave_sum_u32(bad_count);
        if ((tid & 63) == 0) {
            if (nl_count) atomicAdd(&h.rec_info[1], (u64)nl_count);
            if (bad_count) atomicAdd(&h.rec_info[2], (u64)bad_count);
        }
    }
    if (h.acgt) {
        a0 = wave_sum_u32(a0); a1 = wave_sum_u32(a1); a2 = wave_sum_u32(a2); a3 = wave_sum_u32(a3);
        if ((tid & 63) == 0) {
            if (a0) atomicAdd(&h.acgt[0], (u64)a0);
            if (a1) atomicAdd(&h.acgt[1], (u64)a1);
            if (a2) atomicAdd(&h.acgt[2], (u64)a2);
            if (a3) atomicAdd(&h.acgt[3], (u64)a3);
        }
    }
}

// offs[w][d] = ghist0[d] (exclusive prefix over the digits, already in place) + rows[0 .. w)[d]
__global__ void rows_scan_kernel(const u32* __restrict__ rows, const u64* __restrict__ ghist0, u32 ranges, u32 radix, u64* __restrict__ offs) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= radix) return;
    u64 run = ghist0[d];
    for (u32 w = 0; w < ranges; w++) {
        offs[(u64)w * radix + d] = run;
        run += rows[(u64)w * radix + d];
    }
}

// ---------------------------------------------------------------------------------------
// the pass
// ---------------------------------------------------------------------------------------
struct P0Args {
    const u32* gcodes;    // the stream's 2-bit image and validity bits, one word / half word per 16 bytes (the histogram kernel wrote them)
    const u16* gvalid;
    u64 n_bytes;
    int K;
    StreamTiling tl;
    u32 ranges;           // stream ranges; the grid is twice that for ZK_KEYS_BOTH: workgroup ranges + w writes the reverse strand of range w
    int by_record;        // the host has seen the histogram kernel's verdict: threads follow the records (else positions)
    int shift, bits;
    const u64* offs;      // [ranges][1 << bits]
    const u32* rows;      // [ranges][1 << bits]: checked against what the pass wrote
    u64* kout;
    u16* dig_out;         // PLANES: the next pass's digit of every key, (key >> dig_shift) & dig_mask (kout then holds the keys' low words, u32)
    int dig_shift;
    u32 dig_mask;
    u64 n;                // keys in all (every digit's piece ends at or before it; a store never leaves its digit's piece)
    u32* err;
    int split_stores;     // 1: a tile's units leave in two bursts (see the kernel)
    int dbg_mode;         // diagnostic build (-DZK_PHASES) only, measurements (results are wrong): 1 = no stores, 2 = after a range's first tile only the stores (its keys again and again)
    u64* dbg;             // or null (zk_debug_buffer): [ranges][16] time (s_memtime ticks) wave 0 of the range spent per phase, summed over its tiles
};
// phase accounting and measurement modes for tools/p0_phases.py: compiled in only with -DZK_PHASES
#ifdef ZK_PHASES          // make CXXFLAGS_EXTRA=-DZK_PHASES: the diagnostic build (phase accounting and the measurement modes)
#define P0_MODE (a.dbg_mode)
#define P0_PHASE(k) do { if (a.dbg) { const u32 now__ = (u32)__builtin_amdgcn_s_memtime(); ph[k] += now__ - tlast; tlast = now__; } } while (0)
#else
#define P0_MODE 0
#define P0_PHASE(k) do { } while (0)
#endif

// inclusive prefix sum over the 64 lanes with DPP moves (row shifts, then the two row broadcasts of gfx9): no LDS round trips
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);          // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);          // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);          // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);          // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);          // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);          // row_bcast:31 into rows 2 and 3
    return v;
}

// A unit of the output: up to G keys that lie side by side in exch and go to G consecutive, G-aligned places of the output (the
// first unit of a digit's piece may start off the grid, the last may be short).  One word: slot | (keys - 1) << 14 | digit << 18;
// the place in the output is gbase[digit] + slot.
__device__ __forceinline__ u32 unit_pack(u32 slot, u32 len, u32 digit) { return slot | ((len - 1u) << 14) | (digit << 18); }

template <int RBITS, int G>
struct P0Smem {
    static constexpr int RADIX = 1 << RBITS, CAP = P0_TILE + RADIX * (G - 1), UNITS = CAP / G + 2 * RADIX;
    u64 exch[CAP + 64];            // the tile, grouped by digit; what is left of a digit stays here until the next tile
                                   // (+ one slot per lane for the writes of windows that are no k-mers: no branch around an LDS access)
    u64 gbase[RADIX];              // output index of exch slot 0 as seen by this digit
    u32 units[UNITS];              // what leaves with this tile
    u32 codes[P0_IMG_WORDS], valid[P0_IMG_WORDS];          // the 2-bit image of the tile whose keys are made next
    u32 cnt[RADIX + 64];           // keys of the digit: the left-over ones between tiles, all of them after the ranking (+ 64 for the dead)
    u16 off[RADIX];                // where the digit's keys start in exch
    u16 nu[RADIX];                 // units of the digit
    u32 nunits;
    u32 anybad;                    // some digit has more keys than its piece holds (the histogram and the pass disagree): no further stores
};

// One digit per thread (RADIX == P0_BLOCK): the digit's output cursor and its left-over keys' place are that thread's registers.
// CANON: the key is min(x, rc x) (else x).  FAST: 2 K > 32 and the digit lies in the key's high word (the usual plan): no mask on the
// low words, the digit is one bit-field extract.
// PLANES: the keys leave as two arrays -- the low 32 bits (a u32 at kout) and the next pass's digit (a u16 at dig_out), same index --
// instead of whole keys: the 9 bits of this pass's digit are said by the key's place, the next pass reads 6 bytes a key, not 8.
template <int RBITS, int G, bool CANON, bool FAST, bool PLANES = false>
__global__ __launch_bounds__(P0_BLOCK, 4) void stream_pass0_kernel(P0Args a) {
    using S = P0Smem<RBITS, G>;
    constexpr int RADIX = S::RADIX, NW = P0_NW, BLOCK = P0_BLOCK;
    static_assert(S::CAP < (1 << 14) && G <= 16 && RBITS <= 14, "unit_pack");
    static_assert(RADIX == BLOCK && (G & (G - 1)) == 0 && 64 % G == 0, "one digit per thread; whole units per wave instruction");
    __shared__ S sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 w = blockIdx.x;          // the row of offs / rows
    const bool rev = !CANON && w >= a.ranges;          // (ZK_KEYS_BOTH: the second half of the grid)
    const u32 radix = 1u << a.bits;
    const u64 B = (u64)(rev ? w - a.ranges : w) * a.tl.range_bytes;
    if (B >= a.n_bytes) return;
    const u64 E = (B + a.tl.range_bytes < a.n_bytes) ? B + a.tl.range_bytes : a.n_bytes;
    const u32 tile_bytes = a.by_record ? a.tl.tile_bytes : (u32)P0_TILE;
    const u32 ntile = (u32)((E - B + tile_bytes - 1) / tile_bytes);
    const u32 nch = tile_bytes / 16 + 3;          // 16-byte chunks of a tile with its halo
    const int K = a.K;
    const u64 mask = ~0ull >> (64 - 2 * K);
    const u32 mlo = (u32)mask, mhi = (u32)(mask >> 32);
    const u32 dmask = radix - 1u;
    const u32 nchunks = (u32)((a.n_bytes + 15) >> 4);
    auto digit = [&](u64 k) -> u32 {
        if constexpr (FAST) return __builtin_amdgcn_ubfe((u32)(k >> 32), (u32)a.shift - 32u, (u32)a.bits);
        else return (u32)(k >> a.shift) & dmask;
    };
    // this thread's digit
    u64 F = (u32)tid < radix ? a.offs[(u64)w * radix + tid] : 0ull;          // its next output index
    const u64 Fend = F + ((u32)tid < radix ? a.rows[(u64)w * radix + tid] : 0u);       // ... where its piece ends
    u32 nck = 0, ctail = 0;                                                  // keys left over from the last tile, where they sit in exch
    u32 bad = 0;                                                             // more keys of the digit than the histogram had counted
    u32 flen_dbg = 0;
    sm.cnt[tid] = 0;
    if (tid == 0) sm.anybad = 0;
    // this thread's windows inside a tile: NW consecutive ones from tile position p0 on
    u32 p0 = (u32)NW * tid, wlim = (1u << NW) - 1u;          // wlim: which of them exist at all
    if (a.by_record) {
        const u32 r = (u32)(((u64)tid * a.tl.cpr_inv) >> 32), j = tid - r * a.tl.cpr;
        p0 = r * a.tl.rec + (u32)NW * j;
        const int left = (int)a.tl.wpr - NW * (int)j;
        wlim = (r < a.tl.rpt) ? ((left >= NW) ? (1u << NW) - 1u : ((1u << (left > 0 ? left : 0)) - 1u)) : 0u;
        if (r >= a.tl.rpt) p0 = 0;
    }
    // The image of tile t + 1 is made, and the bytes of tile t + 2 are asked for, BEFORE the stores of tile t are issued: a wait for
    // loaded bytes then never has this tile's stores in front of it (the memory counter is one for loads and stores).
    u32 qc = 0, qv = 0;
    auto fetch = [&](u64 T) {          // chunk tid of the tile that starts at stream byte T (a multiple of 16)
        const u64 idx = (T >> 4) + (u32)tid;
        const bool in = (u32)tid < nch && idx < nchunks;
        qc = in ? a.gcodes[idx] : 0u;
        qv = in ? (u32)a.gvalid[idx] : 0u;
    };
    fetch(B);
    if ((u32)tid < nch) { sm.codes[tid] = qc; sm.valid[tid] = qv; }
    if (ntile > 1) fetch(B + tile_bytes);
    __syncthreads();
    u32 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 tlast = a.dbg ? (u32)__builtin_amdgcn_s_memtime() : 0;
    (void)tlast;
    // Units [first, end) of the list out: G / 2 lanes per unit, two keys (16 bytes) per lane, four units of a lane group in flight
    // (their LDS reads are issued together; the values are pinned or the compiler moves each unit's reads back under its own branch).
    struct __attribute__((packed, aligned(8))) Key2 { u64 a, b; };
    auto store_units = [&](u32 first, u32 end, auto nf) {
        constexpr int NF = decltype(nf)::value;          // units of a lane group in flight
        constexpr u32 LPU = G / 2, GROUPS = BLOCK / LPU;
        const u32 j = 2u * ((u32)tid & (LPU - 1)), g0 = (u32)tid / LPU;
        for (u32 base = first; base < end; base += NF * GROUPS) {
            u32 e4[NF];
            u64 k4[NF], l4[NF], b4[NF];
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * GROUPS + g0;
                e4[g] = sm.units[u < (u32)S::UNITS ? u : 0u];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(e4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                k4[g] = sm.exch[(e4[g] & 0x3fffu) + j];
                l4[g] = sm.exch[(e4[g] & 0x3fffu) + j + 1];
                b4[g] = sm.gbase[(e4[g] >> 18) & (u32)(RADIX - 1)];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(k4[g]), "+v"(l4[g]), "+v"(b4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * GROUPS + g0;
                const u32 len1 = (e4[g] >> 14) & 15u;          // keys - 1
                u64* dst = a.kout + (b4[g] + (e4[g] & 0x3fffu) + j);          // (inside the digit's piece by construction)
                if (u < end && P0_MODE != 1) {
                    if (j < len1) { Key2 v; v.a = k4[g]; v.b = l4[g]; *reinterpret_cast<Key2*>(dst) = v; }
                    else if (j == len1) *dst = k4[g];
                }
            }
        }
    };
    // PLANES: the same units as two arrays.  A lane writes 16 bytes here too: half a unit's low words (four tags), or a whole unit's
    // eight next digits -- first every half unit, then every unit.  (Two tags and two digits per lane, the lanes of the whole-key
    // stores: 30.1 ms against 21.3 for whole keys -- twice the store instructions, of 8 and 4 bytes.)
    auto store_units_planes = [&](u32 first, u32 end) {
        constexpr int NF = 4;
        struct __attribute__((packed, aligned(4))) Tag4 { u32 a, b, c, d; };
        struct __attribute__((packed, aligned(2))) Dig8 { u32 a, b, c, d; };
        u32* const tplane = reinterpret_cast<u32*>(a.kout);
        // ---- low words: work item = (unit, half) ----
        for (u32 base = 2 * first; base < 2 * end; base += NF * BLOCK) {
            u32 e4[NF];
            u32 t4[NF][4];
            u64 b4[NF];
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 it = base + g * BLOCK + (u32)tid, u = it >> 1;
                e4[g] = sm.units[u < (u32)S::UNITS ? u : 0u];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(e4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 it = base + g * BLOCK + (u32)tid;
                const u32 sl = (e4[g] & 0x3fffu) + 4u * (it & 1u);
#pragma unroll
                for (int q = 0; q < 4; q++) t4[g][q] = (u32)sm.exch[sl + q];          // (sl + q < CAP + 8: slots past a unit's end are read, never stored)
                b4[g] = sm.gbase[(e4[g] >> 18) & (u32)(RADIX - 1)];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(t4[g][0]), "+v"(t4[g][1]), "+v"(t4[g][2]), "+v"(t4[g][3]), "+v"(b4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 it = base + g * BLOCK + (u32)tid, h4 = 4u * (it & 1u);
                const u32 len = ((e4[g] >> 14) & 15u) + 1u;
                u32* dst = tplane + (b4[g] + (e4[g] & 0x3fffu) + h4);
                if (it < 2 * end && P0_MODE != 1) {
                    if (h4 + 4 <= len) { Tag4 v; v.a = t4[g][0]; v.b = t4[g][1]; v.c = t4[g][2]; v.d = t4[g][3]; *reinterpret_cast<Tag4*>(dst) = v; }
                    else {
#pragma unroll
                        for (int q = 0; q < 3; q++) if (h4 + q < len) dst[q] = t4[g][q];
                    }
                }
            }
        }
        // ---- next digits: work item = unit ----
        for (u32 base = first; base < end; base += NF * BLOCK) {
            u32 e4[NF];
            u32 d8[NF][4];
            u64 b4[NF];
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * BLOCK + (u32)tid;
                e4[g] = sm.units[u < (u32)S::UNITS ? u : 0u];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(e4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 sl = e4[g] & 0x3fffu;
                u32 hw[8];
#pragma unroll
                for (int q = 0; q < 8; q++) hw[q] = (u32)(sm.exch[sl + q] >> 32);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    d8[g][q] = ((hw[2 * q] >> (a.dig_shift - 32)) & a.dig_mask) | (((hw[2 * q + 1] >> (a.dig_shift - 32)) & a.dig_mask) << 16);
                b4[g] = sm.gbase[(e4[g] >> 18) & (u32)(RADIX - 1)];
            }
#pragma unroll
            for (int g = 0; g < NF; g++) asm volatile("" : "+v"(d8[g][0]), "+v"(d8[g][1]), "+v"(d8[g][2]), "+v"(d8[g][3]), "+v"(b4[g]));
#pragma unroll
            for (int g = 0; g < NF; g++) {
                const u32 u = base + g * BLOCK + (u32)tid;
                const u32 len = ((e4[g] >> 14) & 15u) + 1u;
                u16* dst = a.dig_out + (b4[g] + (e4[g] & 0x3fffu));
                if (u < end && P0_MODE != 1) {
                    if (len == 8) { Dig8 v; v.a = d8[g][0]; v.b = d8[g][1]; v.c = d8[g][2]; v.d = d8[g][3]; *reinterpret_cast<Dig8*>(dst) = v; }
                    else {
#pragma unroll
                        for (int q = 0; q < 7; q++) if ((u32)q < len) dst[q] = (u16)(d8[g][q >> 1] >> (16 * (q & 1)));
                    }
                }
            }
        }
    };
    u32 held = 0, held_end = 0;          // units [held, held_end) of the last tile are still to be stored
    for (u32 t = 0; t < ntile; t++) {
        const u64 T0 = B + (u64)t * tile_bytes;
        const bool last = t + 1 == ntile;
        if (P0_MODE == 2 && t > 1) {
            // measurement: the first tile's units again, further down the output
            __syncthreads();
            sm.gbase[tid] += flen_dbg;
            __syncthreads();
        } else {
        // ---- this thread's keys ---------------------------------------------------------------------
        u64 key[NW];
        u32 live;
        {
            const u32 j = p0 >> 4, s = p0 & 15u;
            const u64 sa = ((u64)sm.codes[j] << 32) | sm.codes[j + 1];
            const u64 sb = ((u64)sm.codes[j + 2] << 32) | sm.codes[j + 3];
            u64 v = ((u64)sm.valid[j] << 48) | ((u64)sm.valid[j + 1] << 32) | ((u64)sm.valid[j + 2] << 16) | (u64)sm.valid[j + 3];
            v = runs_of_k(v, K);
            // bit i: window i is a k-mer (of this range: windows from E on belong to the next one)
            live = (__brev((u32)((v << s) >> (64 - NW))) >> (32 - NW)) & wlim;
            if (T0 + p0 >= E) live = 0;
            WindowWords<NW> ww;
            ww.init(sa, sb, s, K);
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const u64 x = ((u64)(ww.xhi(i) & mhi) << 32) | (FAST ? ww.xlo(i) : (ww.xlo(i) & mlo));
                if constexpr (CANON) {
                    const u64 xb = ((u64)(ww.bhi(i) & mhi) << 32) | (FAST ? ww.blo(i) : (ww.blo(i) & mlo));
                    key[i] = x < xb ? x : xb;
                } else {
                    const u64 xb = ((u64)(ww.bhi(i) & mhi) << 32) | (FAST ? ww.blo(i) : (ww.blo(i) & mlo));
                    key[i] = rev ? xb : x;
                }
            }
        }
        // ---- what the last tile left of this thread's digit: into registers, the park below moves it ----------
        // (every LDS access below is unconditional -- a dead window goes to a slot / a counter of its lane's own: under a branch
        // per window the accesses run one after the other, each a full LDS round trip)
        u64 ck[G - 1];
#pragma unroll
        for (int j = 0; j < G - 1; j++) ck[j] = sm.exch[ctail + j];          // ctail + j < CAP + G: inside exch
        // ---- rank: the digit's counter hands out the places (no order to keep) ------------------------------
        u32 rk[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            rk[i] = atomicAdd(&sm.cnt[((live >> i) & 1u) ? digit(key[i]) : (u32)RADIX + (u32)lane], 1u);
        }
        if constexpr (!PLANES) { if (held < held_end) store_units(held, held_end, std::integral_constant<int, 2>()); }          // (the tile's keys and ranks are live here: two in flight; PLANES never holds units back)
        held = held_end = 0;
        P0_PHASE(0);          // keys made, ranks asked for (and the second half of the last tile's units out)
        __syncthreads();
        P0_PHASE(1);          // ... waiting for the other waves
        // ---- per digit: how many keys leave now, where the digit sits in LDS --------------------------------
        // Every wave adds up all the counters by itself (lane l: digits 8 l .. 8 l + 7), so that no wave waits for another's sum.
        u32 O, tot, flen, head, nu;
        {
            const uint4 c0 = reinterpret_cast<const uint4*>(sm.cnt)[2 * lane], c1 = reinterpret_cast<const uint4*>(sm.cnt)[2 * lane + 1];
            const u32 s8 = c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w;
            const u32 inc8 = wave_incl_scan_dpp(s8);
            const u32 before = wave ? (u32)__builtin_amdgcn_readlane((int)inc8, 8 * wave - 1) : 0u;          // digits below 64 * wave
            tot = sm.cnt[tid];
            const u32 inc = wave_incl_scan_dpp(tot);
            O = before + inc - tot;
            u64 end = F + tot;
            if (end > Fend) { bad = 1; sm.anybad = 1; end = Fend > F ? Fend : F; }          // never beyond the digit's piece (then the error word is set)
            u64 Eo = last ? end : (end & ~(u64)(G - 1));
            if (Eo < F) Eo = F;
            flen = (u32)(Eo - F);
            // the keys that leave, cut at the output's grid: a head up to the first grid line (only until the digit is on the grid),
            // whole units, and on the range's last tile a short tail
            head = (u32)(F & (G - 1)) ? G - (u32)(F & (G - 1)) : 0u;
            if (head > flen) head = flen;
            nu = (head ? 1u : 0u) + (flen - head + G - 1) / G;
            sm.nu[tid] = (u16)nu;
        }
        __syncthreads();          // every wave has read the counters; the unit counts are there
        {
            const uint4 c0 = reinterpret_cast<const uint4*>(sm.nu)[lane];          // eight 16-bit counts: digits 8 l .. 8 l + 7
            const u32 s2 = c0.x + c0.y + c0.z + c0.w;                             // two sums side by side (each < 2^16)
            const u32 s8 = (s2 & 0xffffu) + (s2 >> 16);
            const u32 inc8 = wave_incl_scan_dpp(s8);
            const u32 before = wave ? (u32)__builtin_amdgcn_readlane((int)inc8, 8 * wave - 1) : 0u;
            const u32 inc = wave_incl_scan_dpp(nu);
            u32 U = before + inc - nu;
            if (tid == 0) sm.nunits = (u32)__builtin_amdgcn_readlane((int)inc8, 63);
            sm.off[tid] = (u16)O;
            sm.cnt[tid] = tot - flen;
            sm.gbase[tid] = F - O;
            flen_dbg = flen;
            // this digit's units
            u32 slot = O, left = flen;
            if (head) { if (U < (u32)S::UNITS) sm.units[U] = unit_pack(slot, head, (u32)tid); U++; slot += head; left -= head; }
            while (left) {
                const u32 len = left < (u32)G ? left : (u32)G;
                if (U < (u32)S::UNITS) sm.units[U] = unit_pack(slot, len, (u32)tid);
                U++; slot += len; left -= len;
            }
            F += flen;
        }
        __syncthreads();
        P0_PHASE(2);          // scan over the digits (two barriers)
        // ---- park ----------------------------------------------------------------------------------
        {
            u32 at[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) at[i] = sm.off[digit(key[i])];
            // (a slot at or beyond CAP can only come up when the pass finds more keys of a digit than the histogram counted -- `bad`, the
            // launch ends with ZK_DERR_MISMATCH -- and then goes to the lane's dead slot: nothing is ever written outside exch)
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const u32 slot = at[i] + rk[i];
                sm.exch[(((live >> i) & 1u) && slot < (u32)S::CAP) ? slot : (u32)S::CAP + (u32)lane] = key[i];
            }
#pragma unroll
            for (int j = 0; j < G - 1; j++) sm.exch[((u32)j < nck && O + j < (u32)S::CAP) ? O + j : (u32)S::CAP + (u32)lane] = ck[j];
            nck = tot - flen;
            ctail = O + flen;
        }
        P0_PHASE(3);          // parked
        __syncthreads();
        P0_PHASE(4);          // ... waiting
        // ---- the next tile's image; the bytes of the tile after it ------------------------------------------
        if (!last) {
            if ((u32)tid < nch) { sm.codes[tid] = qc; sm.valid[tid] = qv; }          // (this tile's image is dead since its keys were made)
            if (t + 2 < ntile) fetch(T0 + 2ull * tile_bytes);
        }
        P0_PHASE(5);          // next image (includes the wait for its bytes)
        }
        // ---- whole units out: the first half now, the second after the next tile's keys are made (what they read stays as it is
        // until that tile's scan and park): two bursts of stores per tile instead of one, the other workgroup of the CU fills the gaps
        {
            const u32 nunits = (sm.anybad || sm.nunits > (u32)S::UNITS) ? 0u : sm.nunits;          // (after `bad` the list is not to be trusted: nothing leaves, the launch reports ZK_DERR_MISMATCH)
            const u32 half = (PLANES || P0_MODE == 2 || !a.split_stores) ? nunits : (nunits / 2 + 127u) & ~127u;
            if constexpr (PLANES) store_units_planes(0, half < nunits ? half : nunits);
            else store_units(0, half < nunits ? half : nunits, std::integral_constant<int, 4>());
            held = half < nunits ? half : nunits; held_end = nunits;
        }
        P0_PHASE(6);          // stores issued
        __syncthreads();          // the next image is whole; nobody still reads what the next tile's scan and park rewrite
        P0_PHASE(7);          // ... waiting
    }
    if constexpr (!PLANES) { if (held < held_end) store_units(held, held_end, std::integral_constant<int, 2>()); }
    if (a.dbg && tid == 0) {
        for (int k = 0; k < 8; k++) a.dbg[(u64)w * 16 + k] = (u64)ph[k];
        a.dbg[(u64)w * 16 + 8] = ntile;
    }
    if (P0_MODE) return;
    if (bad || ((u32)tid < radix && (F != Fend || nck != 0))) atomicOr(a.err, ZK_DERR_MISMATCH);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int stream_ranges(zk_ctx* c) { return c->stream_ranges > 0 ? c->stream_ranges : 2 * (c->num_cus > 0 ? c->num_cus : 1); }

// histogram of every pass's digit + acgt + the uniformity check + (optionally) the set-aside blocks, and pass 0's digit counts per
// range.  ghist: [MAX_PASSES][gstride] (zeroed here; exclusive prefixes on return, the key count in *d_n).
int stream_hist(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int mode, const PassPlan& plan, u64* ghist, u32 gstride,
                u64* d_acgt, u64* d_n, u64* rec_info, u64* sample, u32 sample_cap, int sample_shift, u64 sample_value, u32* sample_n,
                void* image_room, uint64_t image_room_bytes, StreamRows* out) {
    SHistArgs h = {};
    h.stream = stream; h.n_bytes = n_bytes; h.K = K; h.mode = mode; h.plan = plan;
    u32 nb = 0;
    bool hi = 2 * K > 32;
    for (int p = 0; p < plan.passes; p++) { h.binbase[p] = nb; nb += 1u << plan.bits[p]; if (plan.shift[p] < 32) hi = false; }
    if (sample && sample_shift < 32) hi = false;
    const bool both = mode == ZK_KEYS_BOTH;
    const u32 strands = both ? 2u : 1u;
    h.revbase = nb;
    if (both) nb += 1u << plan.bits[0];
    h.nbins = nb;
    h.ghist = ghist; h.gstride = gstride;
    h.acgt = d_acgt; h.rec_info = rec_info;
    h.sample = sample; h.sample_n = sample_n; h.sample_cap = sample_cap; h.sample_shift = sample_shift; h.sample_value = sample_value;
    h.ranges = (u32)stream_ranges(c);
    h.split = 4;
    h.dbg_atomics = 2;
#ifdef ZK_PHASES
    if (const char* e = getenv("ZK_HIST_ATOMICS")) h.dbg_atomics = atoi(e);
#endif
    const u32 r0 = 1u << plan.bits[0];
    const uint64_t nchunks = (n_bytes + 15) / 16;
    u32* rows; u64* offs;
    ZK_TRY(arena_alloc(c, sizeof(u32) * (uint64_t)strands * h.ranges * r0, (void**)&rows));
    ZK_TRY(arena_alloc(c, sizeof(u64) * (uint64_t)strands * h.ranges * r0, (void**)&offs));
    // the stream's 2-bit image (3/8 of the stream's size): in the room the caller has for it (the tail of the second sort buffer, idle
    // until pass 1), else from the arena
    const uint64_t codes_bytes = (sizeof(u32) * nchunks + 255) & ~255ull, image_bytes = codes_bytes + ((sizeof(u16) * nchunks + 255) & ~255ull);
    char* img = nullptr;
    if (image_room && image_room_bytes >= image_bytes + 256) img = (char*)(((uintptr_t)image_room + image_room_bytes - image_bytes) & ~(uintptr_t)255);
    else ZK_TRY(arena_alloc(c, image_bytes, (void**)&img));
    h.gcodes = (u32*)img; h.gvalid = (u16*)(img + codes_bytes);
    h.rows = rows;
    ZK_HIP(c, hipMemsetAsync(rows, 0, sizeof(u32) * (uint64_t)strands * h.ranges * r0, c->stream));
    ZK_HIP(c, hipMemsetAsync(ghist, 0, sizeof(u64) * MAX_PASSES * gstride, c->stream));
    if (d_acgt) ZK_HIP(c, hipMemsetAsync(d_acgt, 0, sizeof(u64) * 4, c->stream));
    const u32 grid = h.ranges * h.split;
    prof_begin(c, ZK_PROF_HIST_STREAM, n_bytes);
    const bool two = plan.passes == 2 && !both;
    if (hi && two && mode == ZK_KEYS_CANONICAL) hipLaunchKernelGGL((stream_hist_kernel<true, true, 1>), dim3(grid), dim3(SH_BLOCK), nb * sizeof(u32), c->stream, h);
    else if (hi && two) hipLaunchKernelGGL((stream_hist_kernel<true, true>), dim3(grid), dim3(SH_BLOCK), nb * sizeof(u32), c->stream, h);
    else if (hi) hipLaunchKernelGGL((stream_hist_kernel<true, false>), dim3(grid), dim3(SH_BLOCK), nb * sizeof(u32), c->stream, h);
    else if (two) hipLaunchKernelGGL((stream_hist_kernel<false, true>), dim3(grid), dim3(SH_BLOCK), nb * sizeof(u32), c->stream, h);
    else hipLaunchKernelGGL((stream_hist_kernel<false, false>), dim3(grid), dim3(SH_BLOCK), nb * sizeof(u32), c->stream, h);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    out->rows = rows; out->offs = offs; out->ranges = h.ranges; out->strands = strands; out->radix = r0; out->gcodes = h.gcodes; out->gvalid = h.gvalid;
    return ZK_OK;
}

// after the exclusive prefix of pass 0's digits is in ghist0 and the host has read the verdict on the records
int stream_pass0(zk_ctx* c, uint64_t n_bytes, int K, int mode, int shift, int bits, const u64* ghist0, const StreamRows& rows,
                 uint64_t first_nl, bool uniform, u64* kout, uint64_t n, int variant, const StreamPlanes* planes) {
    if (bits < 1 || bits > 9 || (1u << bits) != rows.radix) return fail(c, ZK_EINTERNAL, "stream_pass0: %d digit bits, rows of %u", bits, rows.radix);
    if ((mode == ZK_KEYS_BOTH) != (rows.strands == 2)) return fail(c, ZK_EINTERNAL, "stream_pass0: rows of %u strands for mode %d", rows.strands, mode);
    const u32 grid = rows.ranges * rows.strands;
    hipLaunchKernelGGL(rows_scan_kernel, dim3((rows.radix + 255) / 256), dim3(256), 0, c->stream, rows.rows, ghist0, grid, rows.radix, rows.offs);
    P0Args a = {};
    a.gcodes = rows.gcodes; a.gvalid = rows.gvalid; a.n_bytes = n_bytes; a.K = K;
    a.tl = make_tiling(n_bytes, first_nl, K, rows.ranges);
    a.ranges = rows.ranges;
    a.by_record = (uniform && a.tl.rec) ? 1 : 0;
    a.shift = shift; a.bits = bits; a.offs = rows.offs; a.rows = rows.rows; a.kout = kout; a.n = n; a.err = c->d_err; a.dbg = c->dbg; a.dbg_mode = variant >> 8; a.split_stores = (variant & 0xff) == 3;          // measured: 10.05 vs 8.73 ms on 20 M reads -- the held-back half sits in the key phase's way
    const bool canon = mode == ZK_KEYS_CANONICAL, fast = 2 * K > 32 && shift >= 32;
    if (planes) {
        // two arrays instead of whole keys (the next pass reads them: stream_pass1): only the usual plan asks for it
        if (!canon || !fast) return fail(c, ZK_EINTERNAL, "stream_pass0: planes need the canonical fast plan");
        a.dig_out = planes->dig; a.dig_shift = planes->shift; a.dig_mask = (1u << planes->bits) - 1u;
        prof_begin(c, ZK_PROF_PASS_STREAM, n_bytes + 6 * n);
        hipLaunchKernelGGL((stream_pass0_kernel<9, 8, true, true, true>), dim3(rows.ranges), dim3(P0_BLOCK), 0, c->stream, a);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
        return ZK_OK;
    }
    prof_begin(c, ZK_PROF_PASS_STREAM, n_bytes + 8 * n);
    if (canon && fast) hipLaunchKernelGGL((stream_pass0_kernel<9, 8, true, true>), dim3(grid), dim3(P0_BLOCK), 0, c->stream, a);
    else if (canon) hipLaunchKernelGGL((stream_pass0_kernel<9, 8, true, false>), dim3(grid), dim3(P0_BLOCK), 0, c->stream, a);
    else if (fast) hipLaunchKernelGGL((stream_pass0_kernel<9, 8, false, true>), dim3(grid), dim3(P0_BLOCK), 0, c->stream, a);
    else hipLaunchKernelGGL((stream_pass0_kernel<9, 8, false, false>), dim3(grid), dim3(P0_BLOCK), 0, c->stream, a);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
