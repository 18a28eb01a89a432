// capi.hip -- the extern "C" surface of libzotk.so (include/zotk.h) over the launchers, plus the
// small kernels that belong to no other file: read packing, synthetic reads, checksums.
#include "internal.hpp"
#include "encode_tile.hpp"

namespace zk {
int kmerize(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int flags, double p, uint64_t seed, u64* out_k, u32* out_c,
            uint64_t cap, zk_kmerize_stats* st);
int merge_many(zk_ctx* c, int k, const u64* const* keys, const void* const* cnts, const uint64_t* ns, u64* out_k, void* out_c,
               int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]);
int widen_counts(zk_ctx* c, const u32* in, u64* out, uint64_t n);
int mirror_expand(zk_ctx* c, const u64* ck, const u32* cc, uint64_t n, int K, u64* out_k, u32* out_c, uint64_t cap, uint64_t* n_out);

// ---- (bases, offsets) -> base stream ----------------------------------------------------------
__global__ void pack_reads_kernel(const u8* __restrict__ bases, const u64* __restrict__ offs, u64 n_reads, u8* __restrict__ out) {
    // one wave per read
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = wave; r < n_reads; r += nw) {
        const u64 b = offs[r], e = offs[r + 1];
        u8* o = out + b + r;
        for (u64 i = lane; i < e - b; i += 64) o[i] = bases[b + i];
        if (lane == 0) o[e - b] = '\n';
    }
}

// ---- synthetic reads (zotmer_amd/synth.py is the specification) --------------------------------
__device__ __forceinline__ u64 mix64(u64 z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct SynthArgs { u64 seed, first, count; int L; u64 genome; u32 sub_thr, n_thr; };
__global__ void synth_kernel(SynthArgs a, u8* __restrict__ out) {
    const u64 s1 = mix64(a.seed + 1), s2 = mix64(a.seed + 2), s3 = mix64(a.seed + 3), s4 = mix64(a.seed + 4),
              s5 = mix64(a.seed + 5), s6 = mix64(a.seed + 6);
    const u64 total = a.count * (u64)(a.L + 1);
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 k = t / (u64)(a.L + 1);
        const int j = (int)(t - k * (u64)(a.L + 1));
        if (j == a.L) { out[t] = '\n'; continue; }
        const u64 i = a.first + k;
        const u64 idx = i * (u64)a.L + (u64)j;
        u64 b;
        const u64 e = mix64(s4 + idx);
        if (a.genome) {
            const u64 start = mix64(s2 + i) % (a.genome - (u64)a.L + 1);
            const u64 strand = mix64(s3 + i) & 1;
            const u64 gpos = strand ? start + (u64)(a.L - 1 - j) : start + (u64)j;
            b = mix64(s1 + gpos) & 3;
            if (strand) b = 3 - b;
            if (a.sub_thr && (u32)(e & 0xFFFFFFFFull) < a.sub_thr) b = (b + 1 + mix64(s5 + idx) % 3) & 3;
        } else {
            b = mix64(s6 + idx) & 3;
        }
        u8 ch = "ACGT"[b];
        if (a.n_thr && (u32)(e >> 32) < a.n_thr) ch = 'N';
        out[t] = ch;
    }
}

// ---- checksums ---------------------------------------------------------------------------------
__global__ void checksum_kernel(const u64* __restrict__ k, const u32* __restrict__ c, u64 n, u64* sums) {
    u64 s0 = 0, s1 = 0, s2 = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 w = c ? (u64)c[i] : 1ull;
        s0 += w; s1 += k[i] * w; s2 += murmer(k[i], 0) * w;
    }
    s0 = wave_sum_u64(s0); s1 = wave_sum_u64(s1); s2 = wave_sum_u64(s2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], s2); }
}

// The checker of the full-size runs (bench.py, tools/verify_k.py) shares NOTHING with the product's encoder
// (encode_tile.hpp: 2-bit tile images, funnel-shift windows, revcomp by bit tricks): a thread walks its own piece of the
// stream byte by byte with the reference's rolling state -- forward k-mer shifted left, reverse complement shifted right,
// a run counter that any non-base byte resets (library/basics.py:303-347) -- after a warm-up of K - 1 bytes before the piece.
constexpr int CS_BLOCK = 256, CS_SEG = 128;
__device__ __forceinline__ u32 cs_code(u32 ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}
__global__ __launch_bounds__(CS_BLOCK) void stream_checksum_kernel(const u8* __restrict__ stream, u64 n_bytes, int K, u64* sums) {
    u64 s0 = 0, s1 = 0, s2 = 0;
    u32 a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const u64 mask = (K < 32) ? ((1ull << (2 * K)) - 1) : ~0ull;
    const int top = 2 * (K - 1);
    for (u64 start = ((u64)blockIdx.x * CS_BLOCK + threadIdx.x) * CS_SEG; start < n_bytes; start += (u64)gridDim.x * CS_BLOCK * CS_SEG) {
        const u64 end = (start + CS_SEG < n_bytes) ? start + CS_SEG : n_bytes;
        u64 p = (start >= (u64)(K - 1)) ? start - (u64)(K - 1) : 0;
        u64 fwd = 0, rev = 0;
        u32 run = 0;
        for (; p < end; p++) {
            const u32 b = cs_code(stream[p]);
            if (b > 3) { fwd = rev = 0; run = 0; continue; }
            fwd = ((fwd << 2) | b) & mask;
            rev = (rev >> 2) | ((u64)(3 - b) << top);
            run++;
            if (run >= (u32)K && p >= start) {
                s0 += 2; s1 += fwd + rev; s2 += murmer(fwd, 0) + murmer(rev, 0);
                const u32 f = (u32)(fwd & 3), r = (u32)(rev & 3);
                a0 += (f == 0) + (r == 0); a1 += (f == 1) + (r == 1); a2 += (f == 2) + (r == 2); a3 += (f == 3) + (r == 3);
            }
        }
    }
    s0 = wave_sum_u64(s0); s1 = wave_sum_u64(s1); s2 = wave_sum_u64(s2);
    const u64 t0 = wave_sum_u64(a0), t1 = wave_sum_u64(a1), t2 = wave_sum_u64(a2), t3 = wave_sum_u64(a3);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], s2);
        atomicAdd(&sums[3], t0); atomicAdd(&sums[4], t1); atomicAdd(&sums[5], t2); atomicAdd(&sums[6], t3);
    }
}

// positions[q] = number of elements of the sorted array that are < queries[q]
__global__ void lower_bound_kernel(const u64* __restrict__ a, u64 n, const u64* __restrict__ q, u32 m, u64* __restrict__ pos) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const u64 x = q[t];
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    pos[t] = lo;
}

static u32 grid_for(zk_ctx* c, u64 n, u64 per_block) {
    u64 g = div_up(n, per_block), mx = (u64)c->num_cus * 16;
    return (u32)(g < mx ? (g ? g : 1) : mx);
}
}  // namespace zk

using namespace zk;

#define ZK_ARGS(c, cond) do { if (!(c)) return ZK_EINVAL; zk::enter(c); if (!(cond)) return zk::fail((c), ZK_EINVAL, "bad argument: %s", #cond); } while (0)

extern "C" {

int zk_pack_reads(zk_ctx* c, const uint8_t* d_bases, const uint64_t* d_offs, uint64_t n_reads, uint8_t* d_stream) {
    ZK_ARGS(c, d_offs && d_stream);
    if (n_reads == 0) return ZK_OK;
    hipLaunchKernelGGL(pack_reads_kernel, dim3(grid_for(c, n_reads, 4)), dim3(256), 0, c->stream, d_bases, (const u64*)d_offs, (u64)n_reads, d_stream);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int zk_encode(zk_ctx* c, const uint8_t* d_stream, uint64_t n_bytes, int K, int both, uint64_t* d_out, uint64_t cap,
              uint64_t* n_out, uint64_t acgt[4]) {
    ZK_ARGS(c, n_out && K >= 1 && K <= 32);
    arena_reset(c);
    return encode_list(c, d_stream, n_bytes, K, both ? 1 : 0, (u64*)d_out, cap, n_out, acgt);
}

int zk_subsample(zk_ctx* c, const uint64_t* d_kmers, uint64_t n, uint64_t seed, double p, uint64_t* d_out, uint64_t cap,
                 uint64_t* n_out) {
    ZK_ARGS(c, n_out);
    arena_reset(c);
    return subsample(c, (const u64*)d_kmers, n, seed, p, (u64*)d_out, cap, n_out);
}

int zk_sort_keys(zk_ctx* c, uint64_t* d_keys, uint64_t n, int key_bits) {
    ZK_ARGS(c, key_bits >= 1 && key_bits <= 64);
    if (n == 0) return ZK_OK;
    arena_reset(c);
    ZK_TRY(arena_require(c, 8 * n + n / 64 + (1 << 20), 8 * n + n / 64 + (1 << 20)));          // (+ histograms, the tile sort's bounds)
    u64 *alt, *res;
    ZK_TRY(arena_alloc(c, 8 * n, (void**)&alt));
    ZK_TRY(sort_keys(c, (u64*)d_keys, alt, n, key_bits, &res));
    if (res != (u64*)d_keys) ZK_HIP(c, hipMemcpyAsync(d_keys, res, 8 * n, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return check_device_error(c);
}

int zk_sort_pairs(zk_ctx* c, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, int key_bits) {
    ZK_ARGS(c, key_bits >= 1 && key_bits <= 64);
    if (n == 0) return ZK_OK;
    arena_reset(c);
    ZK_TRY(arena_require(c, 12 * n + n / 64 + (1 << 20), 12 * n + n / 64 + (1 << 20)));
    u64 *alt, *rk; u32 *valt, *rv;
    ZK_TRY(arena_alloc(c, 8 * n, (void**)&alt));
    ZK_TRY(arena_alloc(c, 4 * n, (void**)&valt));
    ZK_TRY(sort_pairs(c, (u64*)d_keys, alt, d_vals, valt, n, key_bits, &rk, &rv));
    if (rk != (u64*)d_keys) {
        ZK_HIP(c, hipMemcpyAsync(d_keys, rk, 8 * n, hipMemcpyDeviceToDevice, c->stream));
        ZK_HIP(c, hipMemcpyAsync(d_vals, rv, 4 * n, hipMemcpyDeviceToDevice, c->stream));
    }
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return check_device_error(c);
}

int zk_rle(zk_ctx* c, const uint64_t* d_sorted, uint64_t n, uint64_t* d_uniq, uint32_t* d_counts, uint64_t cap, uint64_t* n_unique) {
    ZK_ARGS(c, n_unique);
    arena_reset(c);
    return rle(c, (const u64*)d_sorted, n, (u64*)d_uniq, d_counts, cap, n_unique);
}

int zk_sort_count(zk_ctx* c, uint64_t* d_keys, uint64_t n, int key_bits, uint64_t* d_uniq, uint32_t* d_counts, uint64_t cap,
                  uint64_t* n_unique) {
    ZK_ARGS(c, n_unique && key_bits >= 1 && key_bits <= 64);
    *n_unique = 0;
    if (n == 0) return ZK_OK;
    arena_reset(c);
    ZK_TRY(arena_require(c, 8 * n + n / 64 + (1 << 20), 8 * n + n / 64 + (1 << 20)));
    u64 *alt, *res;
    ZK_TRY(arena_alloc(c, 8 * n, (void**)&alt));
    ZK_TRY(sort_keys(c, (u64*)d_keys, alt, n, key_bits, &res));
    return rle(c, res, n, (u64*)d_uniq, d_counts, cap, n_unique);
}

int zk_kmerize(zk_ctx* c, const uint8_t* d_stream, uint64_t n_bytes, int K, int flags, double p, uint64_t seed,
               uint64_t* d_kmers, uint32_t* d_counts, uint64_t cap, zk_kmerize_stats* stats) {
    ZK_ARGS(c, stats);
    return kmerize(c, d_stream, n_bytes, K, flags, p, seed, (u64*)d_kmers, d_counts, cap, stats);
}

int zk_mirror_expand(zk_ctx* c, const uint64_t* d_ck, const uint32_t* d_cc, uint64_t n, int K, uint64_t* d_kmers, uint32_t* d_counts,
                     uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out && (n == 0 || (d_ck && d_cc && d_kmers && d_counts)));
    return mirror_expand(c, (const u64*)d_ck, d_cc, n, K, (u64*)d_kmers, d_counts, cap, n_out);
}

int zk_hist(zk_ctx* c, const void* d_counts, int count_bits, uint64_t n, uint64_t* vals, uint64_t* freq, uint64_t cap_bins,
            uint64_t* n_bins) {
    ZK_ARGS(c, n_bins && (count_bits == 32 || count_bits == 64));
    arena_reset(c);
    return count_hist(c, d_counts, count_bits, n, vals, freq, cap_bins, n_bins);
}

int zk_widen_counts(zk_ctx* c, const uint32_t* d_in, uint64_t* d_out, uint64_t n) {
    ZK_ARGS(c, true);
    return widen_counts(c, d_in, (u64*)d_out, n);
}

int zk_union_sum(zk_ctx* c, const uint64_t* d_xk, const void* d_xc, uint64_t nx, const uint64_t* d_yk, const void* d_yc,
                 uint64_t ny, uint64_t* d_ok, void* d_oc, int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]) {
    ZK_ARGS(c, n_out && (count_bits == 32 || count_bits == 64));
    arena_reset(c);
    return union_sum(c, (const u64*)d_xk, d_xc, nx, (const u64*)d_yk, d_yc, ny, (u64*)d_ok, d_oc, count_bits, cap, n_out, acgt_w);
}

int zk_merge_n(zk_ctx* c, int k, const uint64_t* const* d_keys, const void* const* d_counts, const uint64_t* ns,
               uint64_t* d_ok, void* d_oc, int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]) {
    ZK_ARGS(c, n_out && k >= 0 && (count_bits == 32 || count_bits == 64));
    return merge_many(c, k, (const u64* const*)d_keys, d_counts, ns, (u64*)d_ok, d_oc, count_bits, cap, n_out, acgt_w);
}

int zk_project_dedupe(zk_ctx* c, const uint64_t* d_kmers, uint64_t n, int shift, uint64_t* d_out, uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out && shift >= 0 && shift < 64);
    arena_reset(c);
    return project_dedupe(c, (const u64*)d_kmers, n, shift, (u64*)d_out, cap, n_out);
}

int zk_project(zk_ctx* c, const uint64_t* d_ref, uint64_t n_ref, const uint64_t* d_kmers, const uint64_t* d_counts, uint64_t n,
               uint64_t* d_ok, uint64_t* d_oc, uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out);
    arena_reset(c);
    return project(c, (const u64*)d_ref, n_ref, (const u64*)d_kmers, (const u64*)d_counts, n, (u64*)d_ok, (u64*)d_oc, cap, n_out);
}

int zk_sample(zk_ctx* c, const uint64_t* d_kmers, const uint64_t* d_counts, uint64_t n, uint64_t seed, double p,
              uint64_t* d_ok, uint64_t* d_oc, uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out);
    arena_reset(c);
    return sample_pairs(c, (const u64*)d_kmers, (const u64*)d_counts, n, seed, p, (u64*)d_ok, (u64*)d_oc, cap, n_out);
}

int zk_split(zk_ctx* c, const uint64_t* d_x, uint64_t nx, const uint64_t* d_y, uint64_t ny, uint64_t abc[3]) {
    ZK_ARGS(c, abc);
    arena_reset(c);
    return intersect_count(c, (const u64*)d_x, nx, (const u64*)d_y, ny, abc);
}

int zk_trim(zk_ctx* c, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n, uint64_t lo, uint64_t hi,
            uint64_t* d_ok, void* d_oc, uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out && (count_bits == 32 || count_bits == 64));
    arena_reset(c);
    return trim(c, (const u64*)d_kmers, d_counts, count_bits, n, lo, hi, (u64*)d_ok, d_oc, cap, n_out);
}

int zk_codec64_encode_dev(zk_ctx* c, const uint64_t* d_vals, uint64_t n, int delta, uint64_t* d_words, uint64_t cap, uint64_t* n_words) {
    ZK_ARGS(c, n_words);
    arena_reset(c);
    return codec_encode(c, (const u64*)d_vals, n, delta, (u64*)d_words, cap, n_words);
}

int zk_codec64_encode_u32_dev(zk_ctx* c, const uint32_t* d_vals, uint64_t n, uint64_t* d_words, uint64_t cap, uint64_t* n_words) {
    ZK_ARGS(c, n_words);
    arena_reset(c);
    return codec_encode_u32(c, (const u32*)d_vals, n, (u64*)d_words, cap, n_words);
}

int zk_codec64_decode_dev(zk_ctx* c, const uint64_t* d_words, uint64_t nw, int delta, uint64_t* d_out, uint64_t cap, uint64_t* n_out) {
    ZK_ARGS(c, n_out);
    arena_reset(c);
    return codec_decode(c, (const u64*)d_words, nw, delta, (u64*)d_out, cap, n_out);
}

int zk_undelta(zk_ctx* c, uint64_t* d_vals, uint64_t n, uint64_t base) {
    ZK_ARGS(c, n == 0 || d_vals);
    if (n == 0) return ZK_OK;
    arena_reset(c);
    ZK_TRY(add_u64(c, (u64*)d_vals, 1, base));
    ZK_TRY(scan64_inclusive(c, (u64*)d_vals, n));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return check_device_error(c);
}

int zk_add_u64(zk_ctx* c, uint64_t* d_vals, uint64_t n, uint64_t x) {
    ZK_ARGS(c, n == 0 || d_vals);
    return add_u64(c, (u64*)d_vals, n, x);
}

int zk_fastq_mask(zk_ctx* c, const uint8_t* d_text, uint64_t n, uint32_t line_phase, uint8_t* d_stream, uint64_t* n_newlines) {
    ZK_ARGS(c, n_newlines && d_stream);
    arena_reset(c);
    return fastq_mask(c, d_text, n, line_phase, d_stream, n_newlines);
}

int zk_lower_bound(zk_ctx* c, const uint64_t* d_sorted, uint64_t n, const uint64_t* queries, uint32_t m, uint64_t* positions) {
    ZK_ARGS(c, (m == 0) || (queries && positions));
    if (m == 0) return ZK_OK;
    arena_reset(c);
    u64 *dq, *dp;
    ZK_TRY(arena_require(c, 16ull * m + 4096, 16ull * m + 4096));
    ZK_TRY(arena_alloc(c, 8ull * m, (void**)&dq));
    ZK_TRY(arena_alloc(c, 8ull * m, (void**)&dp));
    ZK_HIP(c, hipMemcpyAsync(dq, queries, 8ull * m, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(lower_bound_kernel, dim3((m + 63) / 64), dim3(64), 0, c->stream, (const u64*)d_sorted, (u64)n, dq, m, dp);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(positions, dp, 8ull * m, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_synth_reads(zk_ctx* c, uint64_t seed, uint64_t first, uint64_t count, int L, uint64_t genome, uint32_t sub_thr,
                   uint32_t n_thr, uint8_t* d_stream) {
    ZK_ARGS(c, L >= 1 && (genome == 0 || genome >= (uint64_t)L));
    if (count == 0) return ZK_OK;
    SynthArgs a{seed, first, count, L, genome, sub_thr, n_thr};
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(c, count * (uint64_t)(L + 1), 256 * 16)), dim3(256), 0, c->stream, a, d_stream);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

int zk_checksum(zk_ctx* c, const uint64_t* d_kmers, const uint32_t* d_counts, uint64_t n, uint64_t sums[3]) {
    ZK_ARGS(c, sums);
    u64* d = c->d_scalars + 12;
    ZK_HIP(c, hipMemsetAsync(d, 0, 3 * sizeof(u64), c->stream));
    if (n) {
        hipLaunchKernelGGL(checksum_kernel, dim3(grid_for(c, n, 256 * 8)), dim3(256), 0, c->stream, (const u64*)d_kmers, d_counts, (u64)n, d);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 12, d, 3 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) sums[i] = c->h_scalars[12 + i];
    return ZK_OK;
}

int zk_capture_filter(zk_ctx* c, const uint8_t* d_stream, uint64_t n_bytes, int K, const uint64_t* d_baits, uint64_t n_baits,
                      uint8_t* d_out, uint64_t* n_reads, uint64_t* n_kept) {
    ZK_ARGS(c, n_reads && n_kept && d_out && K >= 1 && K <= 32);
    arena_reset(c);
    return capture_filter(c, d_stream, n_bytes, K, (const u64*)d_baits, n_baits, d_out, n_reads, n_kept);
}

int zk_stream_checksum(zk_ctx* c, const uint8_t* d_stream, uint64_t n_bytes, int K, uint64_t sums[7]) {
    ZK_ARGS(c, sums && K >= 1 && K <= 32 && (((uintptr_t)d_stream) & 15) == 0);
    u64* d = c->d_scalars + 12;
    ZK_HIP(c, hipMemsetAsync(d, 0, 7 * sizeof(u64), c->stream));
    if (n_bytes) {
        hipLaunchKernelGGL(stream_checksum_kernel, dim3(grid_for(c, n_bytes, (u64)CS_BLOCK * CS_SEG)), dim3(CS_BLOCK), 0, c->stream, d_stream, (u64)n_bytes, K, d);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 12, d, 7 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 7; i++) sums[i] = c->h_scalars[12 + i];
    return ZK_OK;
}

}  // extern "C"
