// internal.hpp -- host-side context shared by the libzotk translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/zotk.h"
#include "common.hpp"

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    int sort_variant = 3;      // radix-sort geometry for key arrays (zk_tune); 3 = 512 threads x 16 keys, 9-bit digits
    int short_sort = 0;        // zk_kmerize: 1 = sort only the top ~log2(n)+3 bits and finish in the mirror stage (opt-in:
                               // pays off on uncorrelated reads only)
    int packed_pairs = 1;      // (k-mer, count) pairs travel as ONE word (k-mer << s | count) through the key kernel when the counts fit the
                               // spare bits above 2K (pipeline.hip); zk_tune, tests
    int wide_tiles = 1;        // radix sort, default geometry: smaller array passes on 16 K-key tiles, one workgroup per CU (radix_sort.hip V6)
    int early_collapse = 1;    // zk_kmerize (canonical): count the copies before the sort is finished, finish it on (k-mer, count) words
                               // (pipeline.hip): 1 = as early as possible (dedupe_kernel after the passes that leave blocks of <= 32 K keys,
                               // else collapse_kernel one pass before the copies are neighbours), 3 = collapse_kernel only,
                               // 2 = a run-length pass of its own once the copies are neighbours (the round's first form), 0 = off
    int side_div = 8;          // ... side list capacity = n / side_div (tests shrink it to force the fallback)
    int pairs_variant = 7;     // ... for (key, u32) pairs: 7 = the pipeline with a payload (1024 threads x 8 pairs), 2 = one workgroup per tile, one serial chain per digit
    int stream_pass = 1;       // the first sort pass (from the base stream): 1 = static ranges, whole 64-byte units written from LDS
                               // (stream_pass.hip; 3 = a tile's units leave in two bursts, for measurements), 0 = the look-back pipeline
    int tag_words = 2;         // zk_kmerize, block dedupe after two passes with at most 32 key bits below the blocks: the second pass writes
                               // only those bits, as 32-bit tags (radix_sort.hip); 2 (default) = ... and takes a key's place in its digit's run
                               // from a returning LDS add wherever a tile holds keys of one bucket of the pass before (pass_pipe_kernel VAR 3:
                               // 23.6-24.0 against 24.7 ms); 1 = tags, every tile ranked by ballots; 0 = whole keys
    int dedupe_variant = 0;    // block dedupe at <= 32 tag bits: dedupe2_kernel's variant (dedupe2.hip), -1 = dedupe_kernel alone
    int dedupe_limit = 65536;  // ... blocks of this many keys or more go to dedupe_kernel (16-bit counts in dedupe2_kernel's table)
    int tag_pass = 0;          // ... 1 = the pass that writes them is tag_pass.hip's count / scan / scatter over static segments, reading pass 0's
                               // keys as two arrays (6 bytes a key); 0 (default) = the look-back pipeline over whole keys.  Measured
                               // (profiles/r04/tag_pass_ab.json): the pass 25.0 -> 17.4 + 2.2 ms, but pass 0 21.3 -> 29.8 ms (half-line units)
    int kway = 1;              // zk_merge_n: 1 = up to 16 lists per pass (kway.hip) from 4 Mi pairs on, 2 = always, 0 = the tree of 2-way passes
    int tile_sort = 1;         // sorts of keys that do not repeat: LSD passes over the top bits, then tiles sorted to the end in LDS (tilesort.hip)
    int dedupe_bits = 0;       // tests: > 0 = the block dedupe with this many block bits whatever the input's size (pipeline.hip)
    int stream_ranges = 0;     // ... ranges the stream is cut into (0 = two per CU; tests use a few so that a range has many tiles)

    // workspace arena: a bump allocator reset at the start of every API call
    char* arena = nullptr;
    uint64_t arena_size = 0;
    uint64_t arena_off = 0;

    // second, grow-only region for buffers whose size is known only mid-call (the mirror step of
    // zk_kmerize): keeps the arena from having to be sized for the worst case
    char* aux = nullptr;
    uint64_t aux_size = 0;

    // decoupled look-back state (persistent: cleared once per 31 launches, see common.hpp)
    u64* status = nullptr;
    uint64_t status_words = 0;
    unsigned short* part16 = nullptr;   // radix sort: per-tile digit counts, 16 bits each (radix_sort.hip)
    uint64_t part16_words = 0;
    u32 epoch = 0;
    u32* d_ticket = nullptr;   // monotonically increasing tile ticket
    u32* d_xticket = nullptr;  // radix sort pipeline: one tile counter per XCD, 128 bytes apart (zeroed per launch)
    int num_xcd = 1;           // XCDs seen by a probe launch at zk_create (8 on MI355X in SPX mode)
    int xcd_group = 0;         // consecutive tiles an XCD takes at a time (zk_tune; 0 = one global tile order, the default)
    u32 ticket_base = 0;

    u32* d_err = nullptr;      // device error word
    u64* d_scalars = nullptr;  // 64 device scalars for small results
    u64* h_scalars = nullptr;  // pinned mirror

    std::string last_error;
    u64* dbg = nullptr;        // diagnostic stamp buffer (zk_debug_buffer), normally null

    // multi-GPU: one RCCL communicator per context (comm.hip); world 1 / rank 0 until zk_comm_init
    void* comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    uint64_t comm_chunk_bytes = 0;      // bytes per message and round of zk_all_to_all_v (0 = 256 MiB; zk_tune, tests)
    int comm_self_loop = 0;             // tests: the piece a rank keeps travels through ncclSend / ncclRecv to itself as well, and one rank
                                        // all-reduces through ncclAllReduce (the RCCL calls of comm.hip on a box with one GPU)

    void* ring = nullptr;               // page-locked staging ring + copy stream of the file <-> device paths (ingest.hip)

    // optional per-launch timing with HIP events on this stream (zk_profile_*)
    struct ProfRec { int tag; uint64_t bytes; hipEvent_t a, b; };
    bool profile = false;
    std::vector<ProfRec> prof;
};

namespace zk {

int fail(zk_ctx* c, int code, const char* fmt, ...);
// HIP's current device is per host thread and other code in the process (torch.cuda.set_device, another ctx) may have
// changed it: every C-ABI entry makes the context's device current before it allocates or launches.
static inline void enter(zk_ctx* c) {
    int d = -1;
    if (c && (hipGetDevice(&d) != hipSuccess || d != c->device)) (void)hipSetDevice(c->device);
}
#define ZK_HIP(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return zk::fail((c), ZK_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define ZK_TRY(expr)             \
    do {                         \
        int r__ = (expr);        \
        if (r__ != ZK_OK) return r__; \
    } while (0)

// arena
void arena_reset(zk_ctx* c);
int arena_alloc(zk_ctx* c, uint64_t bytes, void** p);   // 256-byte aligned; grows the arena when idle
int arena_require(zk_ctx* c, uint64_t want, uint64_t must);
int aux_require(zk_ctx* c, uint64_t bytes, char** p);   // grows (free + malloc) when too small
// look-back state for one launch that needs `words` status words and `tiles` tickets
int lookback_begin(zk_ctx* c, uint64_t words, uint32_t tiles, u32* epoch, u32* ticket_base);
// zeroed 16-bit words for one sort pass (grow-only buffer, cleared on the stream)
int part16_begin(zk_ctx* c, uint64_t words, unsigned short** out);
// read and clear the device error word (after a stream sync); maps it to a ZK_E* code
int check_device_error(zk_ctx* c);

// per-launch timing: bracket a launch with prof_begin / prof_end (no-ops unless enabled)
void prof_begin(zk_ctx* c, int tag, uint64_t algorithmic_bytes);
void prof_end(zk_ctx* c);
void prof_add_bytes(zk_ctx* c, int tag, uint64_t bytes);   // to the tag's last record: bytes written that only the launch's result tells

static inline uint64_t div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ---- launchers implemented in the kernel files (all asynchronous on c->stream) -------------
// radix_sort.hip
constexpr int MAX_PASSES = 8;
struct PassPlan {          // the digits of an LSD sort: pass p takes bits [shift[p], shift[p] + bits[p])
    int passes;
    int shift[MAX_PASSES];
    int bits[MAX_PASSES];
};
int sort_workspace_bytes(uint64_t n, bool pairs, uint64_t* bytes);
int sort_keys(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, u64** result);
int sort_pairs(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, u64** rk, u32** rv);
int sort_keys_upper(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, int lo_bit, u64** result, int prof_tag = ZK_PROF_PASS_PACKED);
int sort_pairs_upper(zk_ctx* c, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int key_bits, int lo_bit, u64** rk, u32** rv);
int sort_pairs_mirrored(zk_ctx* c, const u64* src_k, const u32* src_v, u64* keys, u64* alt, u32* vals, u32* valt, uint64_t n, int K,
                        u64** rk, u32** rv, int lo_bit = 0);
int sort_pairs_rbits(zk_ctx* c);
// sort whose first pass generates the keys from a base stream (encode.hip + radix_sort.hip)
// Ask sort_stream to look before it sorts: the histogram kernel sets aside the keys whose bits from `shift` up equal `value` (a
// few whole blocks of the block dedupe: every copy of their k-mers), and if more than max_ratio of them are distinct the sort is
// declined (return 1, nothing sorted) -- the caller then plans for an input that does not repeat its k-mers.
// want_distinct: the other way round -- the sort goes ahead if the set-aside keys are (nearly) all distinct and is declined if they repeat.
struct StreamSample { int shift; uint64_t value; double max_ratio; uint64_t seen = 0, distinct = 0; bool want_distinct = false; };
// Ask sort_stream to let the last of its two passes write the keys' low 32 bits only (the bits [32, 2K) are the ones sorted:
// lo_bit == 32): *result is then a u32 array, and `cuts` says where each block of equal sorted bits starts (see tag_cuts_kernel).
struct StreamTags { bool written = false; u64* cuts = nullptr; uint32_t blocks = 0; };
struct StreamSrc { const u8* stream; uint64_t n_bytes; int K; int mode; int lo_bit; int hi_bit = 0; StreamSample* sample = nullptr;
                   StreamTags* tags = nullptr; };   // mode: ZK_KEYS_*; sort bits [lo_bit, hi_bit) (hi_bit 0 = 2K)
int sort_rbits(zk_ctx* c);
PassPlan sort_plan_upper(zk_ctx* c, int key_bits, int lo_bit);   // the digits sort_keys_upper(_counted) takes for the bits [lo_bit, key_bits)
// digit counts of the passes to come, taken by the kernel that writes the words they will sort (no histogram pass of its own then)
struct MirrorHist { int passes; int shift[4]; int bits[4]; u64* raw; };          // raw: [MAX_PASSES][512], += by the producer
int sort_first_bits(zk_ctx* c, int key_bits, int lo_bit);
struct DedupeResult {
    uint64_t n_out = 0;          // distinct keys
    uint32_t flags = 0;          // bit 0: a table filled up (unusable), bit 1: counts beyond the packed field exist (patched by dedupe_finish)
    uint32_t n_big = 0;
    u64 *cuts = nullptr, *nwords = nullptr, *incl = nullptr, *big = nullptr, *work = nullptr;
    u32* sub = nullptr;          // [chunks][64] or null: how a block's entries split on the 6 bits after the block bits
    uint32_t chunks = 0;
    int pack = 0;
};
int dedupe_pass(zk_ctx* c, const u64* keys, uint64_t n, int key_bits, int b, int pack, u64* work, uint64_t cap, DedupeResult* r,
                uint64_t* n_in = nullptr, uint64_t max_chunks = 0, const u32* tags = nullptr, const u64* tag_cuts = nullptr);
// the keys back from their tags: key = (block number << tag_bits) | tag
int expand_tags(zk_ctx* c, const u32* tags, const u64* cuts, uint32_t blocks, int tag_bits, u64* keys_out, uint64_t first_block = 0, uint64_t n_blocks = 0);
// packed_out: out_k takes the words themselves, (key << pack) | count, and out_c is not written (the caller merges them as they are)
int dedupe_finish(zk_ctx* c, const DedupeResult& r, u64* out_k, u32* out_c, u64* out_m = nullptr, int K = 0, int gbases = 0,
                  u64** mirror_hist = nullptr, int* mirror_group_bits = nullptr, bool packed_out = false);
int sort_keys_upper_counted(zk_ctx* c, u64* keys, u64* alt, uint64_t n, int key_bits, int lo_bit, u64* counted, u64** result);
int collapse_pass(zk_ctx* c, const u64* keys, uint64_t n, int shift, int bits, int pack, u64* out, uint64_t cap, uint64_t* n_out,
                  uint64_t max_tiles = 0);
int sort_stream(zk_ctx* c, const StreamSrc& src, u64* buf_a, u64* buf_b, uint64_t cap, uint64_t* n_keys,
                uint64_t acgt[4], u64** result);
// stream_pass.hip: histogram + first pass over static stream ranges
struct StreamRows { u32* rows = nullptr; u64* offs = nullptr; u32 ranges = 0, radix = 0; u32* gcodes = nullptr; u16* gvalid = nullptr;
                    u32 strands = 1; };          // 2 (ZK_KEYS_BOTH): rows [ranges + w] are the reverse strand of range w
int stream_hist(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int mode, const PassPlan& plan, u64* ghist, u32 gstride,
                u64* d_acgt, u64* d_n, u64* rec_info, u64* sample, u32 sample_cap, int sample_shift, u64 sample_value, u32* sample_n,
                void* image_room, uint64_t image_room_bytes, StreamRows* out);
// planes (or null): the pass writes the keys' low 32 bits (u32[n] at kout) and the next pass's digit, (key >> shift) & (2^bits - 1),
// as u16[n] at dig -- 6 bytes a key instead of 8; stream_pass1 takes them from there
struct StreamPlanes { u16* dig; int shift, bits; };
int stream_pass0(zk_ctx* c, uint64_t n_bytes, int K, int mode, int shift, int bits, const u64* ghist0, const StreamRows& rows,
                 uint64_t first_nl, bool uniform, u64* kout, uint64_t n, int variant, const StreamPlanes* planes = nullptr);
// tag_pass.hip: the second pass of the two-pass plan over pass 0's two arrays; writes the tags ordered by (d1, d0) and the blocks' starts
int stream_pass1(zk_ctx* c, const u32* tags, const u16* dig, uint64_t n, const u64* ghist0, uint32_t radix0, int bits1, u32* tout, u64* cuts);
// tilesort.hip: the lower bits of a sort, tile by tile in LDS
int tile_sort_top_bits(uint64_t n, int key_bits, int rbits);   // the top bits the LSD passes must have sorted first (0 = not worth it)
int tile_sort(zk_ctx* c, u64* keys, u32* vals, uint64_t n, int key_bits, int top, bool* declined);   // in place
// ... and run-length counted in the same kernel: the distinct keys and their counts (uniq may be keys)
int tile_sort_count(zk_ctx* c, const u64* keys, uint64_t n, int key_bits, int top, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_unique,
                    bool* declined);
// select.hip
int trim(zk_ctx* c, const u64* keys, const void* cnts, int cbits, uint64_t n, u64 lo, u64 hi, u64* ok, void* oc,
         uint64_t cap, uint64_t* n_out);
int project_dedupe(zk_ctx* c, const u64* keys, uint64_t n, int shift, u64* out, uint64_t cap, uint64_t* n_out);
int subsample(zk_ctx* c, const u64* keys, uint64_t n, u64 seed, double p, u64* out, uint64_t cap, uint64_t* n_out);
int sample_pairs(zk_ctx* c, const u64* keys, const u64* cnts, uint64_t n, u64 seed, double p, u64* ok, u64* oc, uint64_t cap,
                 uint64_t* n_out);
int subsample_pairs(zk_ctx* c, u64* keys, u32* cnts, uint64_t n, u64 seed, double p, uint64_t* n_out);   // in place
int encode_list(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, int both, u64* out, uint64_t cap, uint64_t* n_out,
                uint64_t acgt[4]);
int capture_filter(zk_ctx* c, const u8* stream, uint64_t n_bytes, int K, const u64* baits, uint64_t n_baits, u8* out,
                   uint64_t* n_reads, uint64_t* n_kept);
int rle_prefix(zk_ctx* c, const u64* sorted, uint64_t n, int pshift, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_main,
               u64* side_k, u32* side_c, uint64_t side_cap, uint64_t* n_side);
int sample_heads(zk_ctx* c, const u64* keys, uint64_t n, uint64_t* sampled, uint64_t* heads);   // adjacent-distinct count of a prefix
int reduce_by_key(zk_ctx* c, const u64* sorted, const u32* w, uint64_t n, u64* uniq, u32* sums, uint64_t cap, uint64_t* n_out, int pack = 0,
                  uint64_t* max_sum = nullptr);
int rle(zk_ctx* c, const u64* sorted, uint64_t n, u64* uniq, u32* counts, uint64_t cap, uint64_t* n_unique, int pack = 0,
        bool* overflow = nullptr);
int count_hist(zk_ctx* c, const void* counts, int count_bits, uint64_t n, uint64_t* vals, uint64_t* freq,
               uint64_t cap_bins, uint64_t* n_bins);
// codec.hip
int codec_decode(zk_ctx* c, const u64* d_words, uint64_t nw, int delta, u64* d_out, uint64_t cap, uint64_t* n_out);
int fastq_mask(zk_ctx* c, const u8* d_text, uint64_t n, uint32_t line_phase, u8* d_out, uint64_t* n_newlines);
int codec_encode(zk_ctx* c, const u64* d_vals, uint64_t n, int delta, u64* d_words, uint64_t cap, uint64_t* n_words);
int codec_encode_u32(zk_ctx* c, const u32* d_vals, uint64_t n, u64* d_words, uint64_t cap, uint64_t* n_words);
int scan64_inclusive(zk_ctx* c, u64* d_v, uint64_t n);   // in place, asynchronous
int add_u64(zk_ctx* c, u64* d_v, uint64_t n, u64 x);     // v[i] += x, asynchronous
// ingest.hip
void ring_destroy(zk_ctx* c);
// partition.hip
int hash_partition(zk_ctx* c, const u64* keys, const void* cnts, int count_bits, uint64_t n, int world, u64 seed, u64* ok, void* oc,
                   uint64_t* offsets);
int checksum_any(zk_ctx* c, const u64* keys, const void* cnts, int count_bits, uint64_t n, uint64_t sums[3]);
// kway.hip: the union-sum of 2 .. 16 sorted lists in one pass
int kway_union_sum(zk_ctx* c, int k, const u64* const* keys, const void* const* cnts, const uint64_t* ns, u64* out_k, void* out_c, int count_bits,
                   uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]);
// setops.hip
// disjoint: the caller knows that no key occurs in both lists (a canonical list and its mirror image at odd K): no tile waits for
// another; a shared key is reported (ZK_EINTERNAL), never merged wrongly in silence
int union_sum(zk_ctx* c, const u64* A, const void* cA, u64 nA, const u64* B, const void* cB, u64 nB, u64* ok, void* oc,
              int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4], bool disjoint = false);
// 32-bit counts; the B side is ONE array of (key << pack) | count words
int union_sum_packed_b(zk_ctx* c, const u64* A, const u32* cA, u64 nA, const u64* Bp, u64 nB, int pack, u64* ok, u32* oc, uint64_t cap,
                       uint64_t* n_out, bool disjoint = false);
int union_sum_packed_ab(zk_ctx* c, const u64* Ap, u64 nA, const u64* Bp, u64 nB, int pack, u64* ok, u32* oc, uint64_t cap, uint64_t* n_out,
                        bool disjoint = false);
int column_sum(zk_ctx* c, const u64* rows, uint64_t n_rows, int cols, u64* out);   // out[c] = sum of rows[r][c]
int project(zk_ctx* c, const u64* ref, u64 n_ref, const u64* B, const u64* cB, u64 nB, u64* ok, u64* oc, uint64_t cap, uint64_t* n_out);
int intersect_count(zk_ctx* c, const u64* A, u64 nA, const u64* B, u64 nB, uint64_t abc[3]);
}  // namespace zk
