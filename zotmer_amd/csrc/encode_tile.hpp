// encode_tile.hpp -- the sliding-window 2-bit encoder, as a per-tile device routine.
//
// Reference semantics: basics.kmersList (zotmer/library/basics.py:303-347) with the base table
// _nuc (:42-46): every position p of a sequence whose K bases p..p+K-1 are all in AaCcGgTtUu
// yields x = the 2K-bit big-endian packing of those bases (A0 C1 G2 T/U3) and, on the other
// strand, xb = rc(x); any other byte kills every window that overlaps it.
//
// Device input is a "base stream": the sequences of a batch, each followed by one byte that is
// not a base ('\n').  A window can then never span two reads, so "window valid" is simply "K
// valid bytes in a row" and no per-read offsets are read on the device.
//
// A tile of T stream positions is staged once: a workgroup loads T + 32 bytes with 16-byte
// coalesced loads, converts them to a 2-bit packed image plus a validity bit image in LDS
// (16 bases per 32-bit word, first base in the top bits), and every window is then ONE
// funnel shift out of that image -- O(1) per window for any lane-to-position mapping, which
// is what lets the radix sort's first pass and the digit histogram generate their keys
// directly from the stream instead of reading an 8-byte-per-key array.
#pragma once
#include "common.hpp"

namespace zk {

// number of 32-bit image words for a tile of T positions: T/16 data chunks, two halo chunks
// (K-1 <= 31 bytes) and one pad word so a window can always read four consecutive words
template <int T> struct TileImage {
    static constexpr int NCH = T / 16 + 3;
    u32 codes[NCH];
    u32 valid[NCH];
};

// the 16 stream bytes at `off` (zero past the end), as four little-endian words
__device__ __forceinline__ uint4 load_chunk16(const u8* __restrict__ stream, u64 n_bytes, u64 off) {
    uint4 q = make_uint4(0, 0, 0, 0);
    if (off + 16 <= n_bytes) {
        q = *reinterpret_cast<const uint4*>(stream + off);
    } else if (off < n_bytes) {
        u32 w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) {
            u32 c = (off + b < n_bytes) ? stream[off + b] : 0u;
            w[b >> 2] |= c << (8 * (b & 3));
        }
        q = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return q;
}

__device__ __forceinline__ void encode_words16(const uint4 q, u32& codes, u32& vmask);

__device__ __forceinline__ void encode_chunk16(const u8* __restrict__ stream, u64 n_bytes, u64 off,
                                               u32& codes, u32& vmask) {
    encode_words16(load_chunk16(stream, n_bytes, off), codes, vmask);
}

__device__ __forceinline__ void encode_words16(const uint4 q, u32& codes, u32& vmask) {
    const u32 w[4] = {q.x, q.y, q.z, q.w};
    u32 cc = 0, vv = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            u32 c = (w[i] >> (8 * b)) & 0xffu;
            u32 t = (c >> 1) & 3u;               // A0 C1 G3 T2 (also lower case, U == T)
            u32 code = t ^ (t >> 1);             // A0 C1 G2 T3
            u32 d = (c | 0x20u) - 0x61u;         // 'a' -> 0
            u32 ok = (d <= 20u) ? ((0x180045u >> d) & 1u) : 0u;   // a c g t u
            cc = (cc << 2) | code;
            vv = (vv << 1) | ok;
        }
    }
    codes = cc;
    vmask = vv;     // 16 bits, first byte in bit 15
}

// Stage positions [t0, t0 + T) (+ halo) of the stream into `img`.  `stream` must be 16-byte
// aligned and t0 a multiple of 16.  Ends with a workgroup barrier.
template <int BLOCK, int T>
__device__ __forceinline__ void stage_tile(const u8* __restrict__ stream, u64 n_bytes, u64 t0, TileImage<T>& img) {
    for (int c = threadIdx.x; c < TileImage<T>::NCH; c += BLOCK) {
        u32 cc, vv;
        encode_chunk16(stream, n_bytes, t0 + 16ull * c, cc, vv);
        img.codes[c] = cc;
        img.valid[c] = vv;
    }
    __syncthreads();
}

// Window at tile-relative position p (0 <= p < T): returns true iff all K bases are valid.
template <int T>
__device__ __forceinline__ bool window_at(const TileImage<T>& img, int p, int K, u64& x) {
    const int j = p >> 4;
    const int s = p & 15;
    const u64 a = ((u64)img.codes[j] << 32) | img.codes[j + 1];
    const u64 b = ((u64)img.codes[j + 2] << 32) | img.codes[j + 3];
    const u64 hi = s ? ((a << (2 * s)) | (b >> (64 - 2 * s))) : a;
    x = hi >> (64 - 2 * K);
    const u64 v = ((u64)img.valid[j] << 48) | ((u64)img.valid[j + 1] << 32) | ((u64)img.valid[j + 2] << 16) |
                  (u64)img.valid[j + 3];
    const u64 need = (K >= 64) ? ~0ull : ((1ull << K) - 1);
    return ((v << s) >> (64 - K)) == need;
}

// The 16 windows that start in chunk j (positions 16j .. 16j+15), forward and reverse complement, from
// eight LDS words instead of eight per window.  With S = the 64 bases from 16j on as a 128-bit number
// (first base on top) and R = the reverse complement of those 64 bases,
//     x_i  = (S >> 2(64 - i - K)) & mask        rc(x_i) = (R >> 2i) & mask
// (K <= 31, so a window never reaches past base 45 of the 64), and window i is valid iff the validity
// bits i .. i+K-1 are all set, which one "run of K ones" mask answers for all 16.
// Returns the 16 validity bits (bit i = window i).
template <int T>
__device__ __forceinline__ u32 windows16(const TileImage<T>& img, int j, int K, u64 (&x)[16], u64 (&xb)[16]) {
    const u64 a = ((u64)img.codes[j] << 32) | img.codes[j + 1];
    const u64 b = ((u64)img.codes[j + 2] << 32) | img.codes[j + 3];
    const u64 rhi = rev_pairs(~b), rlo = rev_pairs(~a);
    const u64 mask = ~0ull >> (64 - 2 * K);         // 1 <= K <= 32
    u64 v = ((u64)img.valid[j] << 48) | ((u64)img.valid[j + 1] << 32) | ((u64)img.valid[j + 2] << 16) | (u64)img.valid[j + 3];
    int have = 1;                                   // v: bit 63-q = position q; make bit 63-q mean "q .. q+K-1 all valid"
    while (2 * have <= K) { v &= v << have; have *= 2; }
    v &= v << (K - have);
    u32 ok = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int sh = 128 - 2 * i - 2 * K;         // 36 .. 126, uniform
        const u64 f = (sh >= 64) ? (a >> ((sh - 64) & 63)) : ((a << ((64 - sh) & 63)) | (b >> (sh & 63)));
        x[i] = f & mask;
        xb[i] = ((i ? ((rlo >> (2 * i)) | (rhi << (64 - 2 * i))) : rlo)) & mask;
        ok |= (u32)((v >> (63 - i)) & 1ull) << i;
    }
    return ok;
}

// The 16 windows that start at an ARBITRARY tile position p0 (record-aligned tiles): the same algebra with the
// in-chunk offset s = p0 & 15 a per-lane value, so the shifts are register shifts instead of constants.
// s + 15 + K <= 15 + 15 + 32 = 62 bases: still inside the 64 bases of four words.
template <int T>
__device__ __forceinline__ u32 windows16_at(const TileImage<T>& img, int p0, int K, u64 (&x)[16], u64 (&xb)[16]) {
    const int j = p0 >> 4, s = p0 & 15;
    const u64 a = ((u64)img.codes[j] << 32) | img.codes[j + 1];
    const u64 b = ((u64)img.codes[j + 2] << 32) | img.codes[j + 3];
    const u64 rhi = rev_pairs(~b), rlo = rev_pairs(~a);
    const u64 mask = ~0ull >> (64 - 2 * K);
    u64 v = ((u64)img.valid[j] << 48) | ((u64)img.valid[j + 1] << 32) | ((u64)img.valid[j + 2] << 16) | (u64)img.valid[j + 3];
    int have = 1;
    while (2 * have <= K) { v &= v << have; have *= 2; }
    v &= v << (K - have);
    v <<= s;                                        // bit 63-i = window s+i
    u32 ok = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int sh = 128 - 2 * (s + i) - 2 * K;   // 4 .. 126, per lane
        const u64 hi_part = a >> ((sh - 64) & 63);
        const u64 lo_part = (a << ((64 - sh) & 63)) | (b >> (sh & 63));
        x[i] = ((sh >= 64) ? hi_part : lo_part) & mask;
        const int r = 2 * (s + i);                  // 0 .. 60
        xb[i] = (r ? ((rlo >> r) | (rhi << ((64 - r) & 63))) : rlo) & mask;
        ok |= (u32)((v >> (63 - i)) & 1ull) << i;
    }
    return ok;
}

// The same, one window at a time, for a consumer that does not keep the 16 keys (the digit histogram):
// eight registers of state instead of sixty-four of results.
template <int T>
struct Windows16 {
    u64 a, b, rlo, rhi, v, mask;
    int K;
    __device__ __forceinline__ void init(const TileImage<T>& img, int j, int K_) {
        K = K_;
        a = ((u64)img.codes[j] << 32) | img.codes[j + 1];
        b = ((u64)img.codes[j + 2] << 32) | img.codes[j + 3];
        rhi = rev_pairs(~b); rlo = rev_pairs(~a);
        mask = ~0ull >> (64 - 2 * K);
        v = ((u64)img.valid[j] << 48) | ((u64)img.valid[j + 1] << 32) | ((u64)img.valid[j + 2] << 16) | (u64)img.valid[j + 3];
        int have = 1;
        while (2 * have <= K) { v &= v << have; have *= 2; }
        v &= v << (K - have);
    }
    // i is a compile-time constant at every call site (unrolled loops)
    __device__ __forceinline__ bool get(int i, u64& x, u64& xb) const {
        const int sh = 128 - 2 * i - 2 * K;
        const u64 f = (sh >= 64) ? (a >> ((sh - 64) & 63)) : ((a << ((64 - sh) & 63)) | (b >> (sh & 63)));
        x = f & mask;
        xb = (i ? ((rlo >> ((2 * i) & 63)) | (rhi << ((64 - 2 * i) & 63))) : rlo) & mask;
        return (v >> (63 - i)) & 1ull;
    }
};

}  // namespace zk
