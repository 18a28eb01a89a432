// codec.hip -- K11 / K12: the on-disk vector codec of the sorted k-mer-set format on the device.
//
// Reference: zotmer/library/codec64.py:42-151 (word = 4-bit tag n + n fields of 60/n bits, first
// value lowest; greedy encoder: longest run of n <= 6 values whose widest member fits 60/n bits) and
// the delta transform around the k-mers (zotmer/library/files.py:85-110).  Streams are byte-exact with
// the reference's (checked against tests/golden raw members).
//
// K11 decode: a word's tag is its value count -> tile prefix by look-back -> every lane unpacks its
//             word to its slot; k-mers then take a 64-bit inclusive scan (undelta).
// K12 encode: the greedy parse is sequential (a word starts where the previous one ended), but a word
//             holds 1..6 values, so a parse entering a chunk can only be in one of 6 states (offset of
//             its first word start).  Every chunk computes its 6-state transition, the states are
//             resolved across chunks and tiles by composing those maps (decoupled look-back on the
//             3-bit state), and only then are words counted, placed and packed.  Exact, not heuristic.
//
// Algorithmic bytes: decode reads 8 B/word and writes 8 B/value (+16 B/value for the undelta pass);
// encode reads 8 B/value (twice: widths, then packing) and writes 8 B/word.
#include "internal.hpp"

namespace zk {

constexpr int CD_BLOCK = 256;
constexpr int CD_NW = CD_BLOCK / 64;

// field width for tag n as the reference's decoder table gives it (codec64.py:28-31), 0 = no such tag
__device__ __forceinline__ int decode_width(int tag) {
    // tags 1..15 -> 60 30 20 15 12 10 8 7 - 6 - 5 - - 4      packed 6 bits each would not fit one word: two tables
    const u64 lo = (60ull) | (30ull << 8) | (20ull << 16) | (15ull << 24) | (12ull << 32) | (10ull << 40) | (8ull << 48) | (7ull << 56);
    const u64 hi = (0ull) | (6ull << 8) | (0ull << 16) | (5ull << 24) | (0ull << 32) | (0ull << 40) | (4ull << 48);
    if (tag >= 1 && tag <= 8) return (int)((lo >> (8 * (tag - 1))) & 0xff);
    if (tag >= 9 && tag <= 15) return (int)((hi >> (8 * (tag - 9))) & 0xff);
    return 0;
}

struct CdState {
    u64* status; u32* ticket; u32 ticket_base; u32 epoch; u32* err; u64* d_total; u32 tiles;
};

struct CdSmem {
    u64 wtot[CD_NW];
    u64 tile_excl;
    u32 ticket;
};

// block-wide exclusive offsets of per-thread amounts in BLOCKED order (thread t before t+1);
// returns this thread's global exclusive offset; total over all tiles goes to *st.d_total (last tile)
__device__ __forceinline__ u64 blocked_offsets(CdSmem& sm, const CdState& st, u32 tile, u64 mine, u64* tile_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 inc = wave_incl_scan_u64(mine);
    if (lane == 63) sm.wtot[wave] = inc;
    __syncthreads();
    u64 wex = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CD_NW; w++) { if (w < wave) wex += sm.wtot[w]; tot += sm.wtot[w]; }
    if (wave == 0) {
        const u64 ex = lookback_exclusive(st.status, tile, tot, st.epoch, st.err);
        if (lane == 0) { sm.tile_excl = ex; if (tile == st.tiles - 1) *st.d_total = ex + tot; }
    }
    __syncthreads();
    *tile_total = tot;
    return sm.tile_excl + wex + inc - mine;
}

// ---------------------------------------------------------------------------------------
// K11 decode
// ---------------------------------------------------------------------------------------
constexpr int DEC_ITEMS = 8;                      // consecutive words per thread
constexpr int DEC_TILE = CD_BLOCK * DEC_ITEMS;

__global__ __launch_bounds__(CD_BLOCK) void decode_kernel(const u64* __restrict__ words, u64 nw, u64* __restrict__ out, u64 cap,
                                                          CdState st) {
    __shared__ CdSmem sm;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const u64 base = (u64)tile * DEC_TILE + (u64)threadIdx.x * DEC_ITEMS;
    u64 w[DEC_ITEMS];
    u32 cnt = 0;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < DEC_ITEMS; i++) {
        w[i] = (base + i < nw) ? words[base + i] : 0ull;
        if (base + i < nw) {
            const int tag = (int)(w[i] & 15);
            if (!decode_width(tag)) bad = true;
            cnt += (u32)tag;
        }
    }
    if (bad) atomicOr(st.err, ZK_DERR_BAD_TAG);
    u64 tile_total;
    u64 off = blocked_offsets(sm, st, tile, cnt, &tile_total);
#pragma unroll
    for (int i = 0; i < DEC_ITEMS; i++) {
        if (base + i < nw) {
            const int tag = (int)(w[i] & 15);
            const int b = decode_width(tag);
            if (b) {
                u64 v = w[i] >> 4;
                const u64 msk = (1ull << b) - 1;
                for (int m = 0; m < tag; m++) {
                    if (off < cap) out[off] = v & msk;
                    v >>= b;
                    off++;
                }
            }
        }
    }
    if (threadIdx.x == 0 && tile == st.tiles - 1 && sm.tile_excl + tile_total > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
}

// 64-bit inclusive prefix sum in place (undelta, files.py:100-110).  The tile carry travels through
// TWO look-back chains (low and high 32 bits of the tile sums), each well inside the 57-bit payload.
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = CD_BLOCK * SCAN_ITEMS;

__global__ __launch_bounds__(CD_BLOCK) void scan64_kernel(u64* __restrict__ v, u64 n, u64* status_lo, u64* status_hi, CdState st) {
    __shared__ CdSmem sm;
    __shared__ u64 carry[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const u64 base = (u64)tile * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u64 x[SCAN_ITEMS];
    u64 mine = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        x[i] = (base + i < n) ? v[base + i] : 0ull;
        mine += x[i];
        x[i] = mine;                                  // inclusive within the thread
    }
    const u64 inc = wave_incl_scan_u64(mine);
    if (lane == 63) sm.wtot[wave] = inc;
    __syncthreads();
    u64 wex = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CD_NW; w++) { if (w < wave) wex += sm.wtot[w]; tot += sm.wtot[w]; }
    if (wave == 0) {
        const u64 lo = lookback_exclusive(status_lo, tile, tot & 0xFFFFFFFFull, st.epoch, st.err);
        if (lane == 0) carry[0] = lo;
    } else if (wave == 1) {
        const u64 hi = lookback_exclusive(status_hi, tile, tot >> 32, st.epoch, st.err);
        if (lane == 0) carry[1] = hi;
    }
    __syncthreads();
    const u64 add = carry[0] + (carry[1] << 32) + wex + inc - mine;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) v[base + i] = x[i] + add;
}

// in-place 64-bit inclusive prefix sum of a device array (asynchronous on the stream)
int scan64_inclusive(zk_ctx* c, u64* d_v, uint64_t n) {
    if (n == 0) return ZK_OK;
    CdState s2;
    s2.tiles = (u32)div_up(n, SCAN_TILE);
    ZK_TRY(lookback_begin(c, 2ull * s2.tiles, s2.tiles, &s2.epoch, &s2.ticket_base));
    s2.status = c->status; s2.ticket = c->d_ticket; s2.err = c->d_err; s2.d_total = c->d_scalars + 9;
    hipLaunchKernelGGL(scan64_kernel, dim3(s2.tiles), dim3(CD_BLOCK), 0, c->stream, d_v, (u64)n, c->status, c->status + s2.tiles, s2);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

__global__ void add_u64_kernel(u64* __restrict__ v, u64 n, u64 x) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) v[i] += x;
}

// v[i] += x for i < n (asynchronous)
int add_u64(zk_ctx* c, u64* d_v, uint64_t n, u64 x) {
    if (n == 0 || x == 0) return ZK_OK;
    u64 g = div_up(n, 256 * 8), mx = (u64)c->num_cus * 16;
    hipLaunchKernelGGL(add_u64_kernel, dim3((u32)(g < mx ? g : mx)), dim3(256), 0, c->stream, d_v, (u64)n, x);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int codec_decode(zk_ctx* c, const u64* d_words, uint64_t nw, int delta, u64* d_out, uint64_t cap, uint64_t* n_out) {
    *n_out = 0;
    if (nw == 0) return ZK_OK;
    CdState st;
    st.tiles = (u32)div_up(nw, DEC_TILE);
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    hipLaunchKernelGGL(decode_kernel, dim3(st.tiles), dim3(CD_BLOCK), 0, c->stream, d_words, (u64)nw, d_out, (u64)cap, st);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t n = c->h_scalars[9];
    *n_out = n;                      // also when the output was too small: lets the caller size it
    ZK_TRY(check_device_error(c));
    if (delta && n) {
        ZK_TRY(scan64_inclusive(c, d_out, n));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        ZK_TRY(check_device_error(c));
    }
    return ZK_OK;
}

// ---------------------------------------------------------------------------------------
// K12 encode
// ---------------------------------------------------------------------------------------
constexpr int ENC_CH = 32;                         // positions per thread (one chunk)
constexpr int ENC_TILE = CD_BLOCK * ENC_CH;        // 8192 positions per tile

template <class T>
__device__ __forceinline__ u64 enc_value(const T* __restrict__ v, u64 i, int delta) {
    return delta ? (i ? (u64)v[i] - (u64)v[i - 1] : (u64)v[i]) : (u64)v[i];
}
__device__ __forceinline__ int bit_length64(u64 x) { return x ? 64 - __builtin_clzll(x) : 0; }

// greedy word length if a word started at position i: 0 marks a value with no code (>= 2^60)
template <class T>
__global__ void enc_len_kernel(const T* __restrict__ v, u64 n, int delta, u8* __restrict__ len, u32* err) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        int cnt = 0, mw = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (cnt == k && i + k < n) {
                const int w = bit_length64(enc_value(v, i + k, delta));
                const int mwx = w > mw ? w : mw;
                if (mwx <= 60 / (k + 1)) { mw = mwx; cnt = k + 1; }
            }
        }
        if (cnt == 0) { atomicOr(err, ZK_DERR_RANGE); cnt = 1; }    // keep the parse moving; the call fails afterwards
        len[i] = (u8)cnt;
    }
}

// a 6-state map packed 3 bits per entry: bits [3e, 3e+3) = exit offset for entry offset e
__device__ __forceinline__ u32 map_apply(u32 m, u32 s) { return (m >> (3 * s)) & 7u; }
__device__ __forceinline__ u32 map_compose(u32 first, u32 then) {       // s -> then(first(s))
    u32 r = 0;
#pragma unroll
    for (int e = 0; e < 6; e++) r |= map_apply(then, map_apply(first, (u32)e)) << (3 * e);
    return r;
}
constexpr u32 MAP_ID = 0 | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15);

struct EncSmem {
    CdSmem cd;
    u8 len[ENC_TILE + 8];
    u32 wmap[CD_NW];
    u32 entry;            // true entry offset of the tile
};

// Look-back on the parse state.  status word value: PARTIAL = the tile's 18-bit map, INCLUSIVE = the
// exit offset (3 bits) of the tile under the true parse.  Called by one thread.
__device__ __forceinline__ u32 lookback_state(u64* status, u32 tile, u32 my_map, u32 epoch, u32* err) {
    if (tile == 0) {
        const u32 ex = map_apply(my_map, 0);
        st_agent(&status[0], st_pack(ZK_ST_INCLUSIVE, epoch, ex));
        return 0;
    }
    st_agent(&status[tile], st_pack(ZK_ST_PARTIAL, epoch, my_map));
    u32 acc = MAP_ID;         // composition of the maps of tiles (t-j, t-1], applied after the state found
    u32 entry = 0;
    for (u32 t = tile; t > 0; t--) {
        u64 w = ld_agent(&status[t - 1]);
        int spins = 0;
        while (st_state(w, epoch) == 0) {
            if (++spins > ZK_SPIN_LIMIT) { atomicOr(err, ZK_DERR_SPIN_TIMEOUT | (256u << 8)); break; }
            __builtin_amdgcn_s_sleep(1);
            w = ld_agent(&status[t - 1]);
        }
        if (st_state(w, epoch) == ZK_ST_PARTIAL) acc = map_compose((u32)(w & 0x3FFFF), acc);
        else { entry = map_apply(acc, (u32)(w & 7)); break; }
        if (t == 1) entry = map_apply(acc, 0);       // walked past tile 0 as PARTIAL (cannot happen: tile 0 is INCLUSIVE)
    }
    st_agent(&status[tile], st_pack(ZK_ST_INCLUSIVE, epoch, map_apply(my_map, entry)));
    return entry;
}

template <class T>
__global__ __launch_bounds__(CD_BLOCK) void encode_kernel(const T* __restrict__ v, u64 n, int delta, const u8* __restrict__ len,
                                                          u64* __restrict__ words, u64 cap, u64* status_state, CdState st) {
    __shared__ EncSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.cd.ticket) - st.ticket_base;
    const u64 tile_base = (u64)tile * ENC_TILE;
    const u64 rem = n - tile_base;
    const int tile_len = rem < (u64)ENC_TILE ? (int)rem : ENC_TILE;
    for (int i = tid; i < ENC_TILE + 8; i += CD_BLOCK) sm.len[i] = (i < tile_len) ? len[tile_base + i] : (u8)1;
    __syncthreads();
    // ---- transition of my chunk for each of the 6 entry offsets ---------------------------------
    const int c0 = tid * ENC_CH, c1 = c0 + ENC_CH;
    u32 mymap = 0;
#pragma unroll
    for (int e = 0; e < 6; e++) {
        int p = c0 + e;
        while (p < c1) p += sm.len[p];
        mymap |= (u32)(p - c1) << (3 * e);
    }
    // ---- prefix composition over the chunks of the tile (blocked order) --------------------------
    u32 incl = mymap;                                   // maps of chunks [wave start .. me]
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 prev = __shfl_up(incl, o, 64);
        if (lane >= o) incl = map_compose(prev, incl);
    }
    if (lane == 63) sm.wmap[wave] = incl;
    __syncthreads();
    u32 before = MAP_ID, whole = MAP_ID;                // chunks of earlier waves; the whole tile
#pragma unroll
    for (int w = 0; w < CD_NW; w++) {
        if (w < wave) before = map_compose(before, sm.wmap[w]);
        whole = map_compose(whole, sm.wmap[w]);
    }
    if (tid == 0) sm.entry = lookback_state(status_state, tile, whole, st.epoch, st.err);
    __syncthreads();
    const u32 prev_lane = __shfl_up(incl, 1, 64);
    u32 upto_me = (lane == 0) ? before : map_compose(before, prev_lane);   // everything before my chunk
    const int my_entry = (int)map_apply(upto_me, sm.entry);
    // ---- count my words, place them, pack them ------------------------------------------------------
    u32 nwords = 0;
    for (int p = c0 + my_entry; p < c1 && p < tile_len; p += sm.len[p]) nwords++;
    u64 tile_total;
    u64 off = blocked_offsets(sm.cd, st, tile, nwords, &tile_total);
    for (int p = c0 + my_entry; p < c1 && p < tile_len; p += sm.len[p]) {
        const int cnt = sm.len[p];
        const int b = 60 / cnt;
        u64 w = 0;
        for (int m = cnt - 1; m >= 0; m--) w = (w << b) | enc_value(v, tile_base + p + m, delta);
        if (off < cap) words[off] = (w << 4) | (u64)cnt;
        off++;
    }
    if (tid == 0 && tile == st.tiles - 1 && sm.cd.tile_excl + tile_total > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
}

// T: the values' type -- u64, or u32 (the counts of zot kmerize, encoded without being widened first)
template <class T>
static int codec_encode_t(zk_ctx* c, const T* d_vals, uint64_t n, int delta, u64* d_words, uint64_t cap, uint64_t* n_words) {
    *n_words = 0;
    if (n == 0) return ZK_OK;
    u8* len;
    ZK_TRY(arena_require(c, n + (1 << 20), n + (1 << 20)));
    ZK_TRY(arena_alloc(c, n + 64, (void**)&len));
    u64 g = div_up(n, 256 * 8);
    if (g > (u64)c->num_cus * 16) g = (u64)c->num_cus * 16;
    hipLaunchKernelGGL((enc_len_kernel<T>), dim3((u32)g), dim3(256), 0, c->stream, d_vals, (u64)n, delta, len, c->d_err);
    ZK_HIP(c, hipGetLastError());
    CdState st;
    st.tiles = (u32)div_up(n, ENC_TILE);
    ZK_TRY(lookback_begin(c, 2ull * st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    hipLaunchKernelGGL((encode_kernel<T>), dim3(st.tiles), dim3(CD_BLOCK), 0, c->stream, d_vals, (u64)n, delta, len, d_words, (u64)cap,
                       c->status + st.tiles, st);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_words = c->h_scalars[9];
    return check_device_error(c);
}
int codec_encode(zk_ctx* c, const u64* d_vals, uint64_t n, int delta, u64* d_words, uint64_t cap, uint64_t* n_words) {
    return codec_encode_t<u64>(c, d_vals, n, delta, d_words, cap, n_words);
}
int codec_encode_u32(zk_ctx* c, const u32* d_vals, uint64_t n, u64* d_words, uint64_t cap, uint64_t* n_words) {
    return codec_encode_t<u32>(c, d_vals, n, 0, d_words, cap, n_words);
}

// ---------------------------------------------------------------------------------------
// f2: FASTQ text -> base stream on the device.  file.readFastq (zotmer/library/file.py:38-52) takes
// lines in groups of four and keeps the second; here every byte learns the number of its line (count
// of '\n' before it: per-thread popcount, block scan, look-back over the tiles) and every byte that is
// not on a sequence line becomes '\n'.  The stream keeps the text's length -- header, '+' and quality
// lines turn into separator runs, which the encode kernels skip like any other non-base byte (quality
// strings MUST go: they contain the letters A C G T) -- so nothing is compacted or copied on the host.
// ---------------------------------------------------------------------------------------
constexpr int FQ_BYTES = 16;                        // bytes per thread
constexpr int FQ_TILE = CD_BLOCK * FQ_BYTES;

__global__ __launch_bounds__(CD_BLOCK) void fastq_mask_kernel(const u8* __restrict__ text, u64 n, u32 phase0, u8* __restrict__ out,
                                                              CdState st) {
    __shared__ CdSmem sm;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const u64 base = (u64)tile * FQ_TILE + (u64)threadIdx.x * FQ_BYTES;
    u8 b[FQ_BYTES];
    if (base + FQ_BYTES <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(text + base);
        const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < FQ_BYTES; i++) b[i] = (u8)(w[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < FQ_BYTES; i++) b[i] = (base + i < n) ? text[base + i] : (u8)0;
    }
    u32 nl = 0;
#pragma unroll
    for (int i = 0; i < FQ_BYTES; i++) nl += (base + i < n && b[i] == '\n') ? 1u : 0u;
    u64 tile_total;
    u64 line = (u64)phase0 + blocked_offsets(sm, st, tile, nl, &tile_total);      // line number of my first byte
    u8 o[FQ_BYTES];
#pragma unroll
    for (int i = 0; i < FQ_BYTES; i++) {
        const bool is_nl = b[i] == '\n';
        o[i] = ((line & 3) == 1 && !is_nl) ? b[i] : (u8)'\n';
        line += is_nl ? 1 : 0;
    }
    if (base + FQ_BYTES <= n) {
        u32 w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < FQ_BYTES; i++) w[i >> 2] |= (u32)o[i] << (8 * (i & 3));
        *reinterpret_cast<uint4*>(out + base) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
#pragma unroll
        for (int i = 0; i < FQ_BYTES; i++) if (base + i < n) out[base + i] = o[i];
    }
}

int fastq_mask(zk_ctx* c, const u8* d_text, uint64_t n, uint32_t line_phase, u8* d_out, uint64_t* n_newlines) {
    *n_newlines = 0;
    if (n == 0) return ZK_OK;
    if (((uintptr_t)d_text & 15) || ((uintptr_t)d_out & 15)) return fail(c, ZK_EINVAL, "text buffers must be 16-byte aligned");
    CdState st;
    st.tiles = (u32)div_up(n, FQ_TILE);
    ZK_TRY(lookback_begin(c, st.tiles, st.tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    hipLaunchKernelGGL(fastq_mask_kernel, dim3(st.tiles), dim3(CD_BLOCK), 0, c->stream, d_text, (u64)n, line_phase & 3u, d_out, st);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 9, c->d_scalars + 9, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_newlines = c->h_scalars[9];
    return check_device_error(c);
}

}  // namespace zk
