// pipeline.hip -- the device-resident `zot kmerize` batch (zotmer/commands/kmerize.py:450-562)
// and the k-way merge of `zot merge` (zotmer/commands/merge.py:165-253).
//
// kmerize, reference: reads(..., both=True) -> kmersList emits x AND rc(x) for every valid
// window (commands/kmerize.py:490, library/reads.py:113-114); KmerAccumulator2 buffers them,
// radix-sorts and run-length counts (:370-437).  The result is strand-symmetric:
// count(x) == count(rc x), and a palindrome (x == rc x, even K only) is counted twice per window.
//
// Here (default, ZK_KMERIZE_CANONICAL): sort only ONE key per window, c = min(x, rc x) -- half
// the sort volume --, run-length count it, then rebuild both strands exactly: the pairs
// (rc c, n) are sorted by key and union-summed with (c, n); a palindrome meets itself there and
// gets n + n, which is what two emissions per window give.  Any deterministic representative
// would do because it never leaves the device.  ZK_KMERIZE_BOTH sorts both strands directly
// (the literal reference path; kept as a cross-check and for tests).
// -D (murmer subsample, :494-509) is a per-VALUE predicate, so it is applied to the counted set
// instead of to every instance; acgt is taken before any filtering, as in the reference (:492-493).
#include <string.h>

#include <vector>

#include "internal.hpp"

namespace zk {

__global__ void mirror_kernel(const u64* __restrict__ c, const u32* __restrict__ n, u64 m, int K, u64* __restrict__ r,
                              u32* __restrict__ v) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        r[i] = revcomp(K, c[i]);
        v[i] = n[i];
    }
}

// ---- the mirror list by groups (see kmerize_full) ---------------------------------------------------------
constexpr int MIRROR_GROUP_BITS = 18;      // 9 bases: 2^18 groups, tables of 4 MB
constexpr int MIRROR_GROUP_BASES = MIRROR_GROUP_BITS / 2;

// start[g] = first index whose top gbits bits are >= g (g = 2^gbits: n)
__global__ void mirror_bounds_kernel(const u64* __restrict__ c, u64 n, int K, int gbits, u64* __restrict__ start) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g > (1u << gbits)) return;
    const u64 want = (u64)g << (2 * K - gbits);
    u64 lo = 0, hi = n;
    if (g == (1u << gbits)) lo = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (c[mid] < want) lo = mid + 1; else hi = mid;
    }
    start[g] = lo;
}

// size_v[v] = size of the group whose mirrored keys end in v, i.e. group g = rc(v) (gbits / 2 bases)
__global__ void mirror_sizes_kernel(const u64* __restrict__ start, int gbits, u64* __restrict__ size_v) {
    const u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= (1u << gbits)) return;
    const u32 g = (u32)revcomp(gbits / 2, (u64)v);
    size_v[v] = start[g + 1] - start[g];
}

// place[g] = (where the mirrored words of group g go: the sizes before v = rc(g), inclusive scan minus the group's own) - start[g]:
// the word of entry i of the list goes to place[g] + i.  Indexed by g, the order the copy walks the list in.
__global__ void mirror_place_kernel(const u64* __restrict__ start, const u64* __restrict__ incl_v, int gbits, u64* __restrict__ place) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (1u << gbits)) return;
    const u32 v = (u32)revcomp(gbits / 2, (u64)g);
    place[g] = incl_v[v] - (start[g + 1] - start[g]) - start[g];
}

// pack > 0: r[pos] = (rc(c) << pack) | n -- one word per pair (v is not written); the caller has checked that every n fits.
// mh: the digit counts of the passes that will sort the words above their group bits, taken on the way (it writes every word anyway).
__global__ __launch_bounds__(256) void mirror_copy_kernel(const u64* __restrict__ c, const u32* __restrict__ n, u64 m, int K, int gbits,
                                                          const u64* __restrict__ place, u64* __restrict__ r, u32* __restrict__ v, int pack, MirrorHist mh) {
    __shared__ u32 bins[4 * 512];
    const bool hist = pack && mh.passes > 0;
    if (hist) {
        for (int q = threadIdx.x; q < 4 * 512; q += blockDim.x) bins[q] = 0;
        __syncthreads();
    }
    const int sh = 2 * K - gbits;
    // four entries of a thread in flight at a time
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x; i0 < m; i0 += 4 * stride) {
        u64 x4[4];
        u32 n4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 i = i0 + q * stride;
            x4[q] = i < m ? c[i] : 0;
            n4[q] = i < m ? n[i] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u64 i = i0 + q * stride;
            if (i >= m) break;
            const u64 x = x4[q];
            const u64 pos = place[(u32)(x >> sh)] + i;
            if (pack) {
                const u64 mw = (revcomp(K, x) << pack) | (u64)n4[q];
                r[pos] = mw;
                if (hist) {
#pragma unroll
                    for (int p = 0; p < 4; p++)
                        if (p < mh.passes) atomicAdd(&bins[p * 512 + ((u32)(mw >> mh.shift[p]) & ((1u << mh.bits[p]) - 1u))], 1u);
                }
            } else { r[pos] = revcomp(K, x); v[pos] = n4[q]; }
        }
    }
    if (hist) {
        __syncthreads();
        for (int q = threadIdx.x; q < mh.passes * 512; q += blockDim.x)
            if (bins[q]) atomicAdd(&mh.raw[q], (u64)bins[q]);
    }
}

__global__ void widen_kernel(const u32* __restrict__ in, u64* __restrict__ out, u64 m) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}

static u32 ew_grid(zk_ctx* c, u64 n) {
    u64 g = div_up(n, 256);
    u64 mx = (u64)c->num_cus * 16;
    return (u32)(g < mx ? (g ? g : 1) : mx);
}

int widen_counts(zk_ctx* c, const u32* in, u64* out, uint64_t n) {
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(widen_kernel, dim3(ew_grid(c, n)), dim3(256), 0, c->stream, in, out, (u64)n);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

// (key, n) of the side list -> the pair itself and its mirror
__global__ void side_expand_kernel(const u64* __restrict__ k, const u32* __restrict__ n, u64 m, int K, u64* __restrict__ pk,
                                   u32* __restrict__ pv) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (u64)gridDim.x * blockDim.x) {
        const u64 x = k[i];
        const u32 v = n[i];
        pk[2 * i] = x; pv[2 * i] = v;
        pk[2 * i + 1] = revcomp(K, x); pv[2 * i + 1] = v;
    }
}

static int ilog2_ceil(uint64_t x) { int b = 0; while ((1ull << b) < x && b < 63) b++; return b; }

// Spare bits above a 2K-bit k-mer in a 64-bit word: room for the count when pairs travel as one word.  Worth it from 10
// bits up (K <= 27); 0 = keep key and count apart.
static int pack_bits_for(int K) {
    const int spare = 64 - 2 * K;
    return spare >= 10 ? (spare > 31 ? 31 : spare) : 0;
}

__global__ void max_u32_kernel(const u32* __restrict__ v, u64 n, u32* out) {
    u32 m = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) m = v[i] > m ? v[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

static int max_u32(zk_ctx* c, const u32* v, uint64_t n, uint64_t* out) {
    u32* d = (u32*)(c->d_scalars + 26);
    ZK_HIP(c, hipMemsetAsync(d, 0, sizeof(u64), c->stream));
    if (n) {
        hipLaunchKernelGGL(max_u32_kernel, dim3(ew_grid(c, n)), dim3(256), 0, c->stream, v, (u64)n, d);
        ZK_HIP(c, hipGetLastError());
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 26, c->d_scalars + 26, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *out = c->h_scalars[26] & 0xffffffffull;
    return ZK_OK;
}

// Both strands from the counted canonical list (c, n), c ascending: the pairs (rc c, n) are sorted by key and union-summed
// with (c, n); a palindrome (c == rc c, even K) meets itself there and gets n + n -- two emissions per window, as the
// reference has them (commands/kmerize.py:490, library/reads.py:113-114).  rk / rk2 (8 bytes per entry) and rv / rv2 (4) are
// work buffers; start / dest tables come from the arena.
// pack > 0 (the caller knows every count is below 2^pack, and 2K + pack <= 64): the mirrored pairs travel as single words
// (rc c << pack | n) through the key kernel -- 16 bytes per pair and pass instead of 24 -- and the union reads them so.
static int mirror_union(zk_ctx* c, const u64* sorted, const u32* cnt, uint64_t uc, int K, u64* rk, u64* rk2, u32* rv, u32* rv2,
                        u64* out_k, u32* out_c, uint64_t cap, uint64_t* n_out, int pack = 0) {
    u64* sk; u32* sv;
    if (!(2 * K >= MIRROR_GROUP_BITS + 8 && uc >= (1ull << 16))) pack = 0;
    // As pairs (K >= 28, or counts too large for the field) and long enough: the mirrored keys are all different, so the tile sort
    // applies -- three passes over the top bits, reverse-complemented on load, then the rest in LDS -- instead of the grouping copy
    // and five passes.  Should a tile decline (it cannot on distinct keys unless they crowd under one prefix), the passes do it all.
    const int ttop = (!pack && c->tile_sort) ? tile_sort_top_bits(uc, 2 * K, sort_pairs_rbits(c)) : 0;
    if (ttop) {
        bool declined = false;
        ZK_TRY(sort_pairs_mirrored(c, sorted, cnt, rk, rk2, rv, rv2, uc, K, &sk, &sv, 2 * K - ttop));
        ZK_TRY(tile_sort(c, sk, sv, uc, 2 * K, ttop, &declined));
        if (declined) ZK_TRY(sort_pairs_mirrored(c, sorted, cnt, rk, rk2, rv, rv2, uc, K, &sk, &sv));
    } else
    if (2 * K >= MIRROR_GROUP_BITS + 8 && uc >= (1ull << 16)) {
        // The list (c, n) is sorted by c, so the k-mers that share their first 9 bases are contiguous -- and those are
        // exactly the mirrored keys rc(c) that share their LAST 9 bases, i.e. their low 18 bits.  The first two passes of
        // an LSD sort of the mirrored keys would only move these 2^18 groups around whole; one copy does it: group
        // boundaries by binary search, group order = order of the reversed-complemented prefix, then only the bits
        // above 18 are sorted.
        // As single words, and when the list is long enough for the tables to be small beside it: grouped by the first TWELVE bases
        // (2^24 groups; their bounds are 2^24 binary searches) -- one pass less over the words.
        int gbits = MIRROR_GROUP_BITS;
        if (pack && 2 * K - 24 >= 8 && uc >= (1ull << 26) && c->arena_size - c->arena_off > 3 * (8ull << 24) + (64ull << 20) + uc / 16) gbits = 24;
        const u32 groups = 1u << gbits;
        u64 *start, *incl, *place;
        ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)groups + 1), (void**)&start));
        ZK_TRY(arena_alloc(c, sizeof(u64) * groups, (void**)&incl));
        ZK_TRY(arena_alloc(c, sizeof(u64) * groups, (void**)&place));
        // the digit counts of the passes that will sort the words above their group bits: taken by the copy (sort_keys_upper_counted)
        MirrorHist mh = {};
        if (pack && c->sort_variant == 3) {
            const PassPlan plan = sort_plan_upper(c, 2 * K + pack, gbits + pack);
            if (plan.passes <= 4 && sort_rbits(c) == 9) {
                ZK_TRY(arena_alloc(c, sizeof(u64) * MAX_PASSES * 512, (void**)&mh.raw));
                ZK_HIP(c, hipMemsetAsync(mh.raw, 0, sizeof(u64) * MAX_PASSES * 512, c->stream));
                mh.passes = plan.passes;
                for (int p = 0; p < plan.passes; p++) { mh.shift[p] = plan.shift[p]; mh.bits[p] = plan.bits[p]; }
            }
        }
        prof_begin(c, ZK_PROF_MIRROR, (pack ? 20 : 24) * uc);
        hipLaunchKernelGGL(mirror_bounds_kernel, dim3((groups + 256) / 256), dim3(256), 0, c->stream, sorted, (u64)uc, K, gbits, start);
        hipLaunchKernelGGL(mirror_sizes_kernel, dim3(groups / 256), dim3(256), 0, c->stream, (const u64*)start, gbits, incl);
        ZK_TRY(scan64_inclusive(c, incl, groups));
        hipLaunchKernelGGL(mirror_place_kernel, dim3(groups / 256), dim3(256), 0, c->stream, (const u64*)start, (const u64*)incl, gbits, place);
        hipLaunchKernelGGL(mirror_copy_kernel, dim3(ew_grid(c, uc)), dim3(256), 0, c->stream, sorted, cnt, (u64)uc, K, gbits, (const u64*)place, rk, rv, pack, mh);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
        if (pack) {
            if (mh.passes) ZK_TRY(sort_keys_upper_counted(c, rk, rk2, uc, 2 * K + pack, gbits + pack, mh.raw, &sk));
            else ZK_TRY(sort_keys_upper(c, rk, rk2, uc, 2 * K + pack, gbits + pack, &sk));
            return union_sum_packed_b(c, sorted, cnt, uc, sk, uc, pack, out_k, out_c, cap, n_out, (K & 1) != 0);
        }
        ZK_TRY(sort_pairs_upper(c, rk, rk2, rv, rv2, uc, 2 * K, MIRROR_GROUP_BITS, &sk, &sv));
    } else {
        // small inputs / short k-mers: the histogram and the first pass of the sort read (c, n) and reverse-complement on load
        ZK_TRY(sort_pairs_mirrored(c, sorted, cnt, rk, rk2, rv, rv2, uc, K, &sk, &sv));
    }
    // (odd K: no k-mer is its own reverse complement, and a canonical k-mer's mirror image is not canonical: the two lists share no key)
    return union_sum(c, sorted, cnt, uc, sk, sv, uc, out_k, out_c, 32, cap, n_out, nullptr, (K & 1) != 0);
}

// zk_mirror_expand: the strands of an already counted canonical list (multi-GPU: after the exchange)
int mirror_expand(zk_ctx* c, const u64* ck, const u32* cc, uint64_t n, int K, u64* out_k, u32* out_c, uint64_t cap, uint64_t* n_out) {
    *n_out = 0;
    if (K < 1 || K > 32) return fail(c, ZK_EINVAL, "K must be in 1..32 (got %d)", K);
    if (n == 0) return ZK_OK;
    arena_reset(c);
    const uint64_t a8 = (8 * n + 255) & ~255ull, a4 = (4 * n + 255) & ~255ull;
    int pack = 0;
    if (c->packed_pairs && pack_bits_for(K)) {
        uint64_t mx = 0;
        ZK_TRY(max_u32(c, cc, n, &mx));
        if (mx < (1ull << pack_bits_for(K))) pack = pack_bits_for(K);
    }
    // the mirrored pairs as single words (every count fits beside its k-mer): two word buffers, 16 bytes an entry; as pairs: 24
    const bool words = pack && 2 * K >= MIRROR_GROUP_BITS + 8 && n >= (1ull << 16);
    const uint64_t wbytes = 2 * a8 + (words ? 0 : 2 * a4);
    const uint64_t need = wbytes + (9 << 20) + n / 8 + (words && n >= (1ull << 26) ? (480ull << 20) : 0);          // work buffers, group tables (2^24 groups of long lists), histograms, merge-path partition
    ZK_TRY(arena_require(c, need, need));
    char* w;
    ZK_TRY(arena_alloc(c, wbytes, (void**)&w));
    ZK_TRY(mirror_union(c, ck, cc, n, K, (u64*)w, (u64*)(w + a8), words ? nullptr : (u32*)(w + 2 * a8), words ? nullptr : (u32*)(w + 2 * a8 + a4), out_k, out_c,
                        cap, n_out, pack));
    return check_device_error(c);
}

// One batch of zk_kmerize.  Canonical mode counts the copies of a k-mer BEFORE the sort is finished -- three ways, tried in
// this order (zk_tune ZK_TUNE_EARLY_COLLAPSE picks one for tests):
//   (1) block dedupe: LSD passes over the TOP bits until the blocks of equal top bits are small, then an LDS hash table per
//       block counts the copies and leaves the block sorted (radix_sort.hip::dedupe_kernel) -- two full-size passes on a
//       50 M-read batch, and the counted list is finished; dedupe_finish also prepares the mirror sort;
//   (2) collapse_kernel: LSD passes over the low bits until the copies are within a tile of each other, the next digit ranked
//       tile by tile with the run lengths counted in LDS, words sorted from that bit up;
//   (3) the form described next: passes over the low bits until the copies are neighbours, a run-length pass, pairs above.
// Then the strands are rebuilt (mirror_union, or straight from the blocks).  ZK_KMERIZE_BOTH: every bit of both strands
// sorted, then RLE (the literal path).
//
// Early collapse (3).  Sequencing reads repeat every k-mer `coverage` times, and an LSD sort drags all those
// copies through every pass.  But after the passes over the low b bits the copies of a k-mer are already NEIGHBOURS as soon
// as 2^b is well above the number of keys (two distinct k-mers rarely share their low b bits), so the run-length count
// can be taken THEN: the remaining passes move (k-mer, count) pairs -- one per distinct k-mer instead of one per copy --
// and a final pass adds up the few k-mers whose copies were interleaved with another k-mer's (reduce_by_key).  Exact
// for any input: collapsing adjacent equal keys and summing equal keys later never loses or invents a count; the data
// only decides how much is saved.  Whether it pays is read off a sample of the partially sorted array (its head holds a
// random subset of the k-mers with all their copies); with little duplication the keys finish the sort as before.
// replan (or null): set to 1, with nothing sorted yet, when the look before the sort says that the reads do not repeat their k-mers
// and both strands are wanted -- then sorting the keys of BOTH strands (twice the keys through three passes and the tile sort, which
// also counts) is less work than the canonical keys, their mirrored list and the union of the two; the caller makes room and calls
// again with both_tiles.  both_tiles: `both`, by the tile-sort plan.
static int kmerize_full(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, bool both, u64* buf_a, u64* buf_b, uint64_t cap_keys,
                        u64* out_k, u32* out_c, uint64_t cap, zk_kmerize_stats* st, uint64_t* n_out, bool canonical_only = false,
                        int* replan = nullptr, bool both_tiles = false) {
    StreamSrc src{stream, n_bytes, K, both ? ZK_KEYS_BOTH : ZK_KEYS_CANONICAL, 0};
    if (both && both_tiles) {
        const int tt = tile_sort_top_bits(2 * n_bytes, 2 * K, sort_rbits(c));
        if (tt) {
            uint64_t n = 0;
            u64* sorted = nullptr;
            src.lo_bit = 2 * K - tt;
            ZK_TRY(sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted));
            st->n_windows = n / 2;
            st->n_instances = n;
            *n_out = 0;
            bool declined = false;
            ZK_TRY(tile_sort_count(c, sorted, n, 2 * K, tt, out_k, out_c, cap, n_out, &declined));
            if (declined) {
                u64* res = nullptr;
                ZK_TRY(sort_keys_upper(c, sorted, sorted == buf_a ? buf_b : buf_a, n, 2 * K, 0, &res, ZK_PROF_PASS_KEYS));
                ZK_TRY(rle(c, res, n, out_k, out_c, cap, n_out));
            }
            st->n_canonical = (K & 1) ? *n_out / 2 : 0;          // (odd K: no k-mer is its own reverse complement, the table is two lists of equal length)
            return ZK_OK;
        }
    }
    // low bits to sort before looking for runs: 2^b >= 8 x keys, a whole number of passes, and at least one pass left over
    int collapse_bit = 0;
    int fused_bit = 0;            // > 0: the low passes stop here; the next digit is ranked tile by tile and counted by collapse_kernel
    const int rb = sort_rbits(c);
    if (!both && c->early_collapse) {
        const int b = rb * ((ilog2_ceil(n_bytes) + 3 + rb - 1) / rb);
        if (b + rb / 2 < 2 * K) collapse_bit = b;
        if (collapse_bit >= 2 * rb && (c->early_collapse == 1 || c->early_collapse == 3) && c->packed_pairs && pack_bits_for(K)) fused_bit = collapse_bit - rb;
    }
    // Better still while the blocks of keys that share b sorted bits are small enough for an LDS hash table (dedupe_kernel: at
    // most 32 K keys per block by the stream's length, at least 2 K so that the copies of a k-mer are spread over many
    // wavefronts): the passes then sort the TOP b bits, the table counts each block and leaves it sorted by the bits below --
    // config 2 is done after TWO passes.
    int dedupe_bit = 0;
    if (fused_bit && c->early_collapse == 1) {
        const int lg = ilog2_ceil(n_bytes);
        const int passes = lg > 15 ? (lg - 15 + rb - 1) / rb : 1;
        const int b = rb * passes;
        if (b < fused_bit + rb && b + rb / 2 < 2 * K && b <= 24 && (n_bytes >> b) >= 2048) dedupe_bit = b;
    }
    // tests (ZK_TUNE_DEDUPE_BITS): the same plan forced on an input of any size -- two passes, tags and blocks of a handful of keys
    // on inputs small enough for the oracle
    if (!both && c->early_collapse == 1 && c->dedupe_bits > 0 && c->packed_pairs && pack_bits_for(K) && c->dedupe_bits % rb == 0 &&
        c->dedupe_bits + rb / 2 < 2 * K && c->dedupe_bits <= 24)
        dedupe_bit = c->dedupe_bits;
    uint64_t n = 0;
    u64* sorted = nullptr;
    bool presampled = false;
    int tile_top = 0;          // > 0: LSD passes over these top bits only, then tile_sort_count
    bool tile_counted = false; // ... which has counted the keys as well (tile_uc distinct ones)
    uint64_t tile_uc = 0;
    StreamTags stags;          // set when the last pass wrote 32-bit tags instead of keys (`sorted` is then a u32 array)
    if (dedupe_bit) {
        // Sorting the top bits first only pays if the blocks can then be counted; an input that does not repeat its k-mers would
        // have to start over.  So the histogram kernel sets aside four whole blocks (prefixes AAATCCTA.: every copy of their
        // k-mers) and the sort is declined when they show little duplication -- at the price of one more histogram run.
        StreamSample smp{2 * K - dedupe_bit + 2, (uint64_t)(0x0D71C8E5u >> (32 - (dedupe_bit - 2))), 0.6};
        src.lo_bit = 2 * K - dedupe_bit; src.hi_bit = 0; src.sample = &smp;
        // at most 32 key bits below the blocks: the last pass may write just those (sort_stream decides; K = 25 after two passes)
        if (c->tag_words && src.lo_bit <= 32) src.tags = &stags;
        const int rc = sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted);
        if (rc < 0) return rc;
        src.sample = nullptr;
        src.tags = nullptr;
        if (rc == 1 && replan && !canonical_only && c->tile_sort && tile_sort_top_bits(2 * n_bytes, 2 * K, rb)) { *replan = 1; return ZK_OK; }
        if (rc == 1) {
            dedupe_bit = 0; src.lo_bit = 0;
            // the input does not repeat its k-mers: no collapse will pay either, every key goes to the end of the sort -- the top bits
            // by passes, the rest tile by tile in LDS (tilesort.hip)
            if (c->tile_sort) tile_top = tile_sort_top_bits(n_bytes, 2 * K, rb);
        }
        presampled = smp.seen >= 4096;          // the look was conclusive: no second one after the passes
    }
    bool top_sorted = false;          // sort_stream has run with the tile sort's plan
    if (!dedupe_bit && !tile_top && !both && c->tile_sort && c->early_collapse && !pack_bits_for(K)) {
        // No block dedupe for this input (K >= 28: no room for a count beside the k-mer).  Reads that repeat
        // their k-mers are collapsed after the low passes (below); reads that do not -- a share of a large genome at low coverage:
        // config 5 -- go the other way: the top bits by passes, the rest tile by tile.  Which it is, a look at the keys under one
        // prefix tells (about 2^15 of them: every copy of their k-mers), taken by the histogram kernel of the plan that is tried first.
        const int tt = tile_sort_top_bits(n_bytes, 2 * K, rb);
        const int lg = ilog2_ceil(n_bytes);
        const int pb = lg - 15 < 2 ? 2 : (lg - 15 > 30 ? 30 : lg - 15);
        if (tt && pb < 2 * K) {
            StreamSample smp{2 * K - pb, (uint64_t)(0x0D71C8E5u >> (32 - pb)), 0.6};
            smp.want_distinct = true;
            src.lo_bit = 2 * K - tt; src.hi_bit = 0; src.sample = &smp;
            const int rc = sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted);
            if (rc < 0) return rc;
            src.sample = nullptr;
            if (rc == 1) src.lo_bit = 0;          // they repeat: the plan below
            else { tile_top = tt; top_sorted = true; }
        }
    }
    if (!dedupe_bit && tile_top) {
        src.lo_bit = 2 * K - tile_top; src.hi_bit = 0;
        if (!top_sorted) ZK_TRY(sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted));
        // ... and counted there: the distinct k-mers leave the tiles with their counts -- straight into the caller's arrays when the
        // canonical list is all that is wanted (the batches of library/engine.py), else over the keys, counts in the other buffer
        bool declined = false;
        u64* other = (sorted == buf_a) ? buf_b : buf_a;
        if (canonical_only) ZK_TRY(tile_sort_count(c, sorted, n, 2 * K, tile_top, out_k, out_c, cap, &tile_uc, &declined));
        else ZK_TRY(tile_sort_count(c, sorted, n, 2 * K, tile_top, sorted, (u32*)other, n, &tile_uc, &declined));
        tile_counted = !declined;
        if (declined) {          // a block of equal top bits too long for a tile: every bit by passes, from where the keys are now
            u64* res = nullptr;
            ZK_TRY(sort_keys_upper(c, sorted, other, n, 2 * K, 0, &res, ZK_PROF_PASS_KEYS));
            sorted = res;
        }
        fused_bit = 0;
        collapse_bit = 0;
    } else if (!dedupe_bit) {
        src.hi_bit = fused_bit ? fused_bit : collapse_bit;
        ZK_TRY(sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted));
    }
    st->n_windows = both ? n / 2 : n;
    st->n_instances = both ? n : 2 * n;
    *n_out = 0;
    if (both) {
        st->n_canonical = 0;
        return rle(c, sorted, n, out_k, out_c, cap, n_out);
    }
    u64* other = (sorted == buf_a) ? buf_b : buf_a;
    u32* cnt = (u32*)other;
    uint64_t uc = 0;
    bool in_aux = false;          // the counted list lives in the aux region (collapse path): both sort buffers are free
    uint64_t max_count = 0;       // largest count of the list, when it came for free (packed reduce)
    bool have_max = false;
    bool canon_packed = false;                       // ... and then the counted canonical list is (k-mer << pack | count) words as well
    u64 *mwords = nullptr, *malt = nullptr;          // the mirrored words, grouped by their low MIRROR_GROUP_BITS, when dedupe_finish wrote them
    u64* mhist = nullptr;                            // ... and the digit counts of the passes that sort them, when it took those as well
    int mgroup = MIRROR_GROUP_BITS;                  // ... the low bits they are grouped by (6 more when the blocks told how they split)
    if (dedupe_bit && n) {
        const int pk = pack_bits_for(K);
        DedupeResult r;
        uint64_t n_in = 0;
        bool done = false;
        // the sample: the leading blocks, about a million keys
        const uint64_t nblocks = 1ull << dedupe_bit, per = n / nblocks + 1;
        const u32* tg = stags.written ? (const u32*)sorted : nullptr;
        if (!presampled) ZK_TRY(dedupe_pass(c, sorted, n, 2 * K, dedupe_bit, pk, other, cap_keys, &r, &n_in, (1u << 20) / per + 4, tg, stags.cuts));
        if (presampled || (!(r.flags & 1) && (double)r.n_out <= 0.6 * (double)n_in)) {
            ZK_TRY(dedupe_pass(c, sorted, n, 2 * K, dedupe_bit, pk, other, cap_keys, &r, nullptr, 0, tg, stags.cuts));
            if (!(r.flags & 1)) {
                uc = r.n_out;
                const uint64_t a8 = (8 * uc + 255) & ~255ull, a4 = (4 * uc + 255) & ~255ull;
                char* aux;
                ZK_TRY(aux_require(c, a8 + a4, &aux));
                // Every count fits the field (the usual case) and both strands are wanted: the copy that closes the gaps also
                // writes the mirrored words grouped by their low 18 bits -- the first stage of the mirror sort (mirror_union's
                // grouping copy) for free; they go over the keys' buffer (the keys are counted, the words are in the other one).
                const bool want_m = !(r.flags & 2) && !canonical_only && c->packed_pairs && dedupe_bit == MIRROR_GROUP_BITS &&
                                    2 * K >= MIRROR_GROUP_BITS + 8;
                // ... and the counted list itself stays in words, (k-mer << pk) | count: nobody but the final union reads it
                ZK_TRY(dedupe_finish(c, r, (u64*)aux, (u32*)(aux + a8), want_m ? sorted : nullptr, K, MIRROR_GROUP_BASES, &mhist, &mgroup, want_m));
                if (want_m) { mwords = sorted; malt = other; canon_packed = true; }
                sorted = (u64*)aux; cnt = (u32*)(aux + a8);
                in_aux = true;
                if (!(r.flags & 2)) { max_count = (1ull << pk) - 1; have_max = true; }          // every count fits the field
                done = true;
            }
        }
        if (!done) {
            // little duplication, or a table filled up: the keys (untouched: the words went to the other buffer) are sorted the
            // long way, all their bits -- the two passes over the top bits were for nothing
            u64* res = nullptr;
            if (tg) {
                // ... made again from their tags first (into the buffer the discarded words are in)
                ZK_TRY(expand_tags(c, tg, stags.cuts, stags.blocks, 2 * K - dedupe_bit, other));
                u64* t = sorted; sorted = other; other = t;
            }
            ZK_TRY(sort_keys(c, sorted, other, n, 2 * K, &res));
            sorted = res;
            other = (sorted == buf_a) ? buf_b : buf_a;
            cnt = (u32*)other;
        }
        fused_bit = 0;
        collapse_bit = 0;
    }
    if (fused_bit && n) {
        // Runs are counted inside the tile-local ranking of the next digit above fused_bit (radix_sort.hip::collapse_kernel):
        // that pass writes one word per run instead of every key, and no pass of its own reads the keys again to count.  Its
        // output is tile-major, so the passes over the words start at fused_bit.  Whether it pays: the same kernel over the
        // first tiles (low bits ascending: a random subset of the k-mers with all their copies).
        const int pk = pack_bits_for(K);
        const int cb = sort_first_bits(c, 2 * K + pk, fused_bit + pk);       // the digit the first pass over the words will use
        const uint64_t sample_tiles = 64, tile_keys = 8192;
        uint64_t us = 0, u1 = 0;
        ZK_TRY(collapse_pass(c, sorted, n, fused_bit, cb, pk, other, cap_keys, &us, sample_tiles));
        const uint64_t m = n < sample_tiles * tile_keys ? n : sample_tiles * tile_keys;
        if ((double)us <= 0.6 * (double)m) {
            ZK_TRY(collapse_pass(c, sorted, n, fused_bit, cb, pk, other, cap_keys, &u1));
            u64* res = nullptr;
            ZK_TRY(sort_keys_upper(c, other, sorted, u1, 2 * K + pk, fused_bit + pk, &res));
            const uint64_t a8 = (8 * u1 + 255) & ~255ull, a4 = (4 * u1 + 255) & ~255ull;
            char* aux;
            ZK_TRY(aux_require(c, a8 + a4, &aux));
            ZK_TRY(reduce_by_key(c, res, nullptr, u1, (u64*)aux, (u32*)(aux + a8), u1, &uc, pk, &max_count));
            sorted = (u64*)aux; cnt = (u32*)(aux + a8);
            in_aux = true;
            have_max = true;
        } else {
            u64* res = nullptr;
            ZK_TRY(sort_keys_upper(c, sorted, other, n, 2 * K, fused_bit, &res, ZK_PROF_PASS_KEYS));      // little duplication: every key to the end
            sorted = res;
            other = (sorted == buf_a) ? buf_b : buf_a;
            cnt = (u32*)other;
        }
        collapse_bit = 0;
    }
    if (collapse_bit && n) {
        uint64_t m = 0, heads = 0;
        ZK_TRY(sample_heads(c, sorted, n, &m, &heads));
        const int pk = c->packed_pairs ? pack_bits_for(K) : 0;
        bool done_packed = false;
        if ((double)heads <= 0.6 * (double)m && pk) {
            // The runs as single words (key << pk | length): written beside the keys (not over them: should a run be longer
            // than 2^pk - 1 the keys are still there and the pair path below takes over), the upper bits sorted by the key
            // kernel, the split runs summed straight from the words.
            const uint64_t cap1 = (8 * cap_keys) / 8;                     // `other` holds cap_keys words
            uint64_t u1 = 0;
            bool ovf = false;
            int rc1 = rle(c, sorted, n, other, nullptr, cap1, &u1, pk, &ovf);
            if (rc1 != ZK_OK && rc1 != ZK_ENOSPC) return rc1;
            if (rc1 == ZK_OK && !ovf) {
                u64* res = nullptr;
                ZK_TRY(sort_keys_upper(c, other, sorted, u1, 2 * K + pk, collapse_bit + pk, &res));
                const uint64_t a8 = (8 * u1 + 255) & ~255ull, a4 = (4 * u1 + 255) & ~255ull;
                char* aux;
                ZK_TRY(aux_require(c, a8 + a4, &aux));
                ZK_TRY(reduce_by_key(c, res, nullptr, u1, (u64*)aux, (u32*)(aux + a8), u1, &uc, pk, &max_count));
                sorted = (u64*)aux; cnt = (u32*)(aux + a8);
                in_aux = true;
                done_packed = true;
                have_max = true;
            }
        }
        if (done_packed) {
        } else if ((double)heads <= 0.6 * (double)m) {
            uint64_t u1 = 0;
            ZK_TRY(rle(c, sorted, n, sorted, cnt, n, &u1));          // in place: runs of adjacent equal keys
            const uint64_t a8 = (8 * u1 + 255) & ~255ull, a4 = (4 * u1 + 255) & ~255ull;
            // the second key / count buffers of the pair passes sit behind the lists in the two sort buffers; should the
            // sample have been too optimistic for that (it never is on reads), they go to the aux region instead
            const bool fits = a8 + a4 + 512 <= 8 * cap_keys;
            char* aux;
            ZK_TRY(aux_require(c, (fits ? 1 : 2) * (a8 + a4), &aux));
            u64* alt = fits ? (u64*)((char*)other + a4) : (u64*)(aux + a8 + a4);
            u32* valt = fits ? (u32*)((char*)sorted + a8) : (u32*)(aux + 2 * a8 + a4);
            u64* sk; u32* sv;
            ZK_TRY(sort_pairs_upper(c, sorted, alt, cnt, valt, u1, 2 * K, collapse_bit, &sk, &sv));
            ZK_TRY(reduce_by_key(c, sk, sv, u1, (u64*)aux, (u32*)(aux + a8), u1, &uc));
            sorted = (u64*)aux; cnt = (u32*)(aux + a8);
            in_aux = true;
        } else {
            u64* res = nullptr;
            ZK_TRY(sort_keys_upper(c, sorted, other, n, 2 * K, collapse_bit, &res, ZK_PROF_PASS_KEYS));      // still every key: the dominant passes
            sorted = res;
            other = (sorted == buf_a) ? buf_b : buf_a;
            cnt = (u32*)other;
            collapse_bit = 0;
        }
    }
    // the caller wants the counted canonical list itself (multi-GPU: it is exchanged before the strands are rebuilt): counted into
    // the caller's arrays where the count is still to be taken
    bool direct = false;
    if (tile_counted) { uc = tile_uc; direct = canonical_only; }
    else if (!in_aux && canonical_only) { ZK_TRY(rle(c, sorted, n, out_k, out_c, cap, &uc)); direct = true; }
    else if (!in_aux) ZK_TRY(rle(c, sorted, n, sorted, cnt, n, &uc));     // in place: sorted[0..uc) = distinct canonical k-mers
    st->n_canonical = uc;
    if (uc == 0) return ZK_OK;
    if (canonical_only) {
        if (uc > cap) return fail(c, ZK_ENOSPC, "output holds %llu entries, the batch has %llu distinct canonical k-mers", (unsigned long long)cap, (unsigned long long)uc);
        if (!direct) {
            ZK_HIP(c, hipMemcpyAsync(out_k, sorted, 8 * uc, hipMemcpyDeviceToDevice, c->stream));
            ZK_HIP(c, hipMemcpyAsync(out_c, cnt, 4 * uc, hipMemcpyDeviceToDevice, c->stream));
        }
        *n_out = uc;
        return ZK_OK;
    }
    if (mwords) {
        const int pk = pack_bits_for(K);
        u64* sk = nullptr;
        if (mhist) ZK_TRY(sort_keys_upper_counted(c, mwords, malt, uc, 2 * K + pk, mgroup + pk, mhist, &sk));
        else ZK_TRY(sort_keys_upper(c, mwords, malt, uc, 2 * K + pk, mgroup + pk, &sk));
        if (canon_packed) return union_sum_packed_ab(c, sorted, uc, sk, uc, pk, out_k, out_c, cap, n_out, (K & 1) != 0);
        return union_sum_packed_b(c, sorted, cnt, uc, sk, uc, pk, out_k, out_c, cap, n_out, (K & 1) != 0);
    }
    const uint64_t a8 = (8 * uc + 255) & ~255ull, a4 = (4 * uc + 255) & ~255ull;
    u64 *rk, *rk2; u32 *rv, *rv2;
    if (in_aux) {
        // both sort buffers are free: the mirror sort works there
        rk = buf_a; rv = (u32*)((char*)buf_a + a8);
        rk2 = buf_b; rv2 = (u32*)((char*)buf_b + a8);
    } else {
        char* aux;
        ZK_TRY(aux_require(c, 2 * a8 + 2 * a4, &aux));
        rk = (u64*)aux; rk2 = (u64*)(aux + a8);
        rv = (u32*)(aux + 2 * a8); rv2 = (u32*)(aux + 2 * a8 + a4);
    }
    int pack = 0;
    if (c->packed_pairs && pack_bits_for(K)) {
        if (!have_max) ZK_TRY(max_u32(c, cnt, uc, &max_count));
        if (max_count < (1ull << pack_bits_for(K))) pack = pack_bits_for(K);
    }
    return mirror_union(c, sorted, cnt, uc, K, rk, rk2, rv, rv2, out_k, out_c, cap, n_out, pack);
}

// The short path: sort only the top T bits of the canonical keys (T ~ log2(n) + 3, a whole number
// of digits), so that nearly every group of equal prefix is a single k-mer already in its final
// place; rle_prefix_kernel writes those to the sorted main list and everything else (mixed groups,
// groups cut by a tile edge) to a side list, which simply joins the strand-mirror pairs in the sort
// they need anyway.  Exact for any input; the data only decides how much goes the long way.
// Returns 1 when the side list would not fit (caller falls back to kmerize_full).
static int kmerize_short(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int lo_bit, u64* buf_a, u64* buf_b,
                         uint64_t cap_keys, u64* out_k, u32* out_c, uint64_t cap, zk_kmerize_stats* st, uint64_t* n_out) {
    StreamSrc src{stream, n_bytes, K, ZK_KEYS_CANONICAL, lo_bit};
    uint64_t n = 0;
    u64* sorted = nullptr;
    ZK_TRY(sort_stream(c, src, buf_a, buf_b, cap_keys, &n, st->acgt, &sorted));
    st->n_windows = n;
    st->n_instances = 2 * n;
    *n_out = 0;
    if (n == 0) return ZK_OK;
    char* other = (char*)((sorted == buf_a) ? buf_b : buf_a);
    u32* cnt = (u32*)other;
    uint64_t side_cap = n / (uint64_t)(c->side_div > 0 ? c->side_div : 8) + 64;
    const uint64_t off_k = (4 * n + 255) & ~255ull;
    if (off_k + 12 * side_cap > 8 * cap_keys) side_cap = (8 * cap_keys - off_k) / 12;
    u64* side_k = (u64*)(other + off_k);
    u32* side_c = (u32*)(other + off_k + 8 * side_cap);
    uint64_t um = 0, ns = 0;
    ZK_TRY(rle_prefix(c, sorted, n, lo_bit, sorted, cnt, n, &um, side_k, side_c, side_cap, &ns));
    if (ns > side_cap) return 1;
    st->n_canonical = um + ns;
    const uint64_t m = um + 2 * ns;
    const uint64_t a8 = (8 * m + 255) & ~255ull, a4 = (4 * m + 255) & ~255ull;
    char* aux;
    ZK_TRY(aux_require(c, (ns ? 3 : 2) * (a8 + a4), &aux));
    u64* pk = (u64*)aux; u64* pk2 = (u64*)(aux + a8);
    u32* pv = (u32*)(aux + 2 * a8); u32* pv2 = (u32*)(aux + 2 * a8 + a4);
    if (um) {
        prof_begin(c, ZK_PROF_MIRROR, 24 * um);
        hipLaunchKernelGGL(mirror_kernel, dim3(ew_grid(c, um)), dim3(256), 0, c->stream, sorted, cnt, (u64)um, K, pk, pv);
        prof_end(c);
        ZK_HIP(c, hipGetLastError());
    }
    if (ns) {
        hipLaunchKernelGGL(side_expand_kernel, dim3(ew_grid(c, ns)), dim3(256), 0, c->stream, side_k, side_c, (u64)ns, K, pk + um, pv + um);
        ZK_HIP(c, hipGetLastError());
    }
    u64* sk; u32* sv;
    ZK_TRY(sort_pairs(c, pk, pk2, pv, pv2, m, 2 * K, &sk, &sv));
    uint64_t mr = m;
    if (ns) {
        u64* rk = (u64*)(aux + 2 * a8 + 2 * a4);
        u32* rv = (u32*)(aux + 3 * a8 + 2 * a4);
        ZK_TRY(reduce_by_key(c, sk, sv, m, rk, rv, m, &mr));
        sk = rk; sv = rv;
    }
    return union_sum(c, sorted, cnt, um, sk, sv, mr, out_k, out_c, 32, cap, n_out, nullptr);
}

int kmerize(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int flags, double p, uint64_t seed, u64* out_k, u32* out_c,
            uint64_t cap, zk_kmerize_stats* st) {
    memset(st, 0, sizeof *st);
    if (K < 1 || K > 32) return fail(c, ZK_EINVAL, "K must be in 1..32 (got %d)", K);
    if (n_bytes == 0) return ZK_OK;
    const bool both = (flags & ZK_KMERIZE_BOTH) != 0;
    const bool canonical_only = (flags & ZK_KMERIZE_CANONICAL_ONLY) != 0;
    if (canonical_only && (both || (flags & ZK_KMERIZE_SUBSAMPLE)))
        return fail(c, ZK_EINVAL, "ZK_KMERIZE_CANONICAL_ONLY cannot be combined with ZK_KMERIZE_BOTH or ZK_KMERIZE_SUBSAMPLE");
    const uint64_t cap_keys = both ? 2 * n_bytes : n_bytes;   // one window per stream byte at most
    arena_reset(c);
    // histograms, merge-path partitions, the mirror group tables (4 MB); from 2^29 stream bytes on (block dedupe with 2^18 blocks) the
    // blocks' tables as well: bounds, sizes, and the 2^24 run places of the mirror grouping (336 MB) -- or, where the blocks were not
    // counted that way, the 2^24 group bounds, ends and places of mirror_union (403 MB)
    const uint64_t slack = (16 << 20) + cap_keys / 16 + ((n_bytes >= (1ull << 29) || c->dedupe_bits >= 18) ? (832ull << 20) : 0);          // (+ the block starts of the tag path: 2 MB; the list of declined blocks: 1 MB)
    ZK_TRY(arena_require(c, 16 * cap_keys + slack, 16 * cap_keys + slack));
    u64 *buf_a, *buf_b;
    ZK_TRY(arena_alloc(c, 8 * cap_keys, (void**)&buf_a));
    ZK_TRY(arena_alloc(c, 8 * cap_keys, (void**)&buf_b));

    uint64_t n_out = 0;
    int rc = 1;
    if (!both && c->short_sort && !canonical_only) {
        const int rb = sort_rbits(c);
        const int T = rb * ((ilog2_ceil(n_bytes) + 3 + rb - 1) / rb);
        if (T < 2 * K) {
            rc = kmerize_short(c, stream, n_bytes, K, 2 * K - T, buf_a, buf_b, cap_keys, out_k, out_c, cap, st, &n_out);
            if (rc < 0) return rc;
            if (rc == 1) st->n_canonical = 0;      // side list overflowed: do it the long way
        }
    }
    if (rc == 1) {
        // Reads that do not repeat their k-mers (the look before the sort says so), both strands wanted: the sort is planned again
        // for the keys of both strands, if the device has the room (twice the sort buffers; the mirror's buffers are not needed then)
        int replan = 0;
        size_t mfree = 0, mtotal = 0;
        const uint64_t need2 = 32 * n_bytes + slack + 2 * n_bytes / 16;
        const bool may = !both && !canonical_only && c->tile_sort && hipMemGetInfo(&mfree, &mtotal) == hipSuccess &&
                         (double)need2 < 0.9 * (double)(mfree + c->arena_size + c->aux_size);
        ZK_TRY(kmerize_full(c, stream, n_bytes, K, both, buf_a, buf_b, cap_keys, out_k, out_c, cap, st, &n_out, canonical_only, may ? &replan : nullptr));
        if (replan) {
            arena_reset(c);
            if (need2 > c->arena_size && (double)need2 >= 0.9 * (double)(mfree + c->arena_size) && c->aux) {
                ZK_HIP(c, hipStreamSynchronize(c->stream));
                ZK_HIP(c, hipFree(c->aux));
                c->aux = nullptr; c->aux_size = 0;
            }
            ZK_TRY(arena_require(c, need2, need2));
            const uint64_t cap2 = 2 * n_bytes;
            ZK_TRY(arena_alloc(c, 8 * cap2, (void**)&buf_a));
            ZK_TRY(arena_alloc(c, 8 * cap2, (void**)&buf_b));
            ZK_TRY(kmerize_full(c, stream, n_bytes, K, true, buf_a, buf_b, cap2, out_k, out_c, cap, st, &n_out, false, nullptr, true));
        }
    }
    if (flags & ZK_KMERIZE_SUBSAMPLE) {
        uint64_t kept = 0;
        ZK_TRY(subsample_pairs(c, out_k, out_c, n_out, seed, p, &kept));
        n_out = kept;
    }
    st->n_unique = n_out;
    return check_device_error(c);
}

// k-way union-sum as a balanced tree of 2-way passes; counts are 32- or 64-bit (count_bits).
// ins: k device arrays (keys, counts, n).  The result lands in (out_k, out_c).
int merge_many(zk_ctx* c, int k, const u64* const* keys, const void* const* cnts, const uint64_t* ns, u64* out_k, void* out_c,
               int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]) {
    const uint64_t cb = (uint64_t)count_bits / 8;
    *n_out = 0;
    if (acgt_w) acgt_w[0] = acgt_w[1] = acgt_w[2] = acgt_w[3] = 0;
    if (k <= 0) return ZK_OK;
    arena_reset(c);
    uint64_t total = 0;
    for (int i = 0; i < k; i++) total += ns[i];
    if (k == 1) {
        // union with the empty set: a plain copy that also yields the count-weighted acgt
        return union_sum(c, keys[0], cnts[0], ns[0], keys[0], cnts[0], 0, out_k, out_c, count_bits, cap, n_out, acgt_w);
    }
    // two ping-pong regions, each able to hold every intermediate list of one level
    const uint64_t slack = 1 << 20;
    const uint64_t rbytes = (8 + cb) * total + 512ull * k;
    const uint64_t need = 2 * rbytes + total / 2 + 8192ull * k + slack;   // + merge-path partitions, acgt rows; the k-way pass: its sample (twice), tile bounds
    ZK_TRY(arena_require(c, need, need));
    struct L { const u64* k; const void* c; uint64_t n; };
    std::vector<L> va(k), vb(k);
    L* cur = va.data();
    L* nxt = vb.data();
    for (int i = 0; i < k; i++) cur[i] = L{keys[i], cnts[i], ns[i]};
    char* region[2];
    ZK_TRY(arena_alloc(c, rbytes, (void**)&region[0]));
    ZK_TRY(arena_alloc(c, rbytes, (void**)&region[1]));
    auto in_regions = [&](const void* q) {
        const char* b = (const char*)q;
        return (b >= region[0] && b < region[0] + rbytes) || (b >= region[1] && b < region[1] + rbytes);
    };
    // Fan-in of a level: up to 16 lists in ONE pass (kway.hip) -- the eight sets of a GPU in BASELINE config 4 are one level, 64 sets
    // two -- or pairs (zk_tune ZK_TUNE_KWAY 0: the tree of 2-way passes)
    // (small inputs keep the tree: the k-way pass sorts a sample with the radix-sort pipeline, eight launches that a few thousand pairs
    // do not pay for; ZK_TUNE_KWAY 2 takes it always -- the tests)
    const int F = (c->kway == 2 || (c->kway == 1 && total >= (1ull << 22))) ? 16 : 2;
    int m = k, level = 0;
    while (m > 1) {
        const bool last = (m <= F);
        char* base = region[level & 1];
        uint64_t off = 0;
        int o = 0;
        for (int i = 0; i < m; i += F) {
            const int g = m - i < F ? m - i : F;          // lists of this group
            if (g == 1) {
                // the odd list sits out this level; if it lives in a ping-pong region the level after next would overwrite it, so
                // move it along with this level's outputs
                L x = cur[i];
                if (in_regions(x.k)) {
                    u64* ok = (u64*)(base + off); off += (8 * x.n + 255) & ~255ull;
                    void* oc = (void*)(base + off); off += (cb * x.n + 255) & ~255ull;
                    ZK_HIP(c, hipMemcpyAsync(ok, x.k, 8 * x.n, hipMemcpyDeviceToDevice, c->stream));
                    ZK_HIP(c, hipMemcpyAsync(oc, x.c, cb * x.n, hipMemcpyDeviceToDevice, c->stream));
                    x.k = ok; x.c = oc;
                }
                nxt[o++] = x;
                continue;
            }
            uint64_t capg = 0;
            for (int j = 0; j < g; j++) capg += cur[i + j].n;
            u64* ok; void* oc; uint64_t capo;
            if (last) { ok = out_k; oc = out_c; capo = cap; }
            else {
                ok = (u64*)(base + off); off += (8 * capg + 255) & ~255ull;
                oc = (void*)(base + off); off += (cb * capg + 255) & ~255ull;
                capo = capg;
            }
            uint64_t no = 0;
            if (g == 2) {
                ZK_TRY(union_sum(c, cur[i].k, cur[i].c, cur[i].n, cur[i + 1].k, cur[i + 1].c, cur[i + 1].n, ok, oc, count_bits, capo, &no,
                                 last ? acgt_w : nullptr));
            } else {
                const u64* gk[16]; const void* gc[16]; uint64_t gn[16];
                for (int j = 0; j < g; j++) { gk[j] = cur[i + j].k; gc[j] = cur[i + j].c; gn[j] = cur[i + j].n; }
                ZK_TRY(kway_union_sum(c, g, gk, gc, gn, ok, oc, count_bits, capo, &no, last ? acgt_w : nullptr));
            }
            nxt[o++] = L{ok, oc, no};
        }
        L* t = cur; cur = nxt; nxt = t;
        m = o;
        level++;
    }
    *n_out = cur[0].n;
    return ZK_OK;
}

}  // namespace zk
