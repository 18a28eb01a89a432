/*
 * zotk.h -- C-ABI of libzotk.so, the MI355X (gfx950) k-mer counting and set-algebra core that
 * sits under `zot kmerize | merge | dist | trim`.
 *
 * The reference (drtconway/zotmer, file:line relative to its root) has no FFI: its commands call
 * pure-Python functions.  Each entry point below names the reference function it stands in for;
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C: pointers and sizes only; no C++ or torch types cross this boundary;
 *   - every function returns ZK_OK (0) or a negative ZK_E* code; zk_last_error(ctx) has the text;
 *   - a zk_ctx is bound to one HIP device and one stream; calls on one ctx are serialised by the
 *     caller; use one ctx per GPU (and per thread);
 *   - pointers named d_* are DEVICE pointers (from zk_alloc, hipMalloc or a torch tensor's
 *     data_ptr()); pointers without the prefix are host memory;
 *   - work is queued on the ctx's stream; a function that returns a host scalar (n_out, acgt...)
 *     has synchronised the stream before returning;
 *   - k-mers are uint64, 2 bits per base, first base in the highest used bits (A0 C1 G2 T/U3),
 *     exactly basics.kmer (zotmer/library/basics.py:48-59).  1 <= K <= 32.
 *   - a "base stream" is the concatenation of the sequences of a batch, each followed by one
 *     byte that is not a base (the host parser keeps the line's '\n'); any byte outside
 *     AaCcGgTtUu ends the windows that touch it, as in basics.kmersList (basics.py:329-339).
 *     Stream pointers must be 16-byte aligned.
 *
 * There is no CPU fallback: without a visible MI355X zk_create returns NULL.
 */
#ifndef ZOTK_H
#define ZOTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_EINVAL (-1)    /* bad argument */
#define ZK_ENOMEM (-2)    /* device memory / workspace exhausted */
#define ZK_EHIP (-3)      /* HIP runtime error */
#define ZK_ENOSPC (-4)    /* an output array is too small (capacity argument) */
#define ZK_EOVERFLOW (-5) /* a count does not fit its type (reference: array('I'), kmerize.py:373-374) */
#define ZK_EINTERNAL (-6) /* device-side failure */
#define ZK_ERANGE (-7)    /* codec64: value >= 2^60 has no code (reference: IndexError, codec64.py:33-40) */

typedef struct zk_ctx zk_ctx;

/* ---- context and buffers ------------------------------------------------------------------ */
zk_ctx* zk_create(int device, uint64_t workspace_bytes);   /* NULL when no such GPU */
void zk_destroy(zk_ctx* ctx);
const char* zk_last_error(zk_ctx* ctx);
int zk_set_stream(zk_ctx* ctx, void* hip_stream);          /* borrow a hipStream_t (NULL: own stream) */
void* zk_get_stream(zk_ctx* ctx);
int zk_sync(zk_ctx* ctx);
int zk_reserve(zk_ctx* ctx, uint64_t workspace_bytes);     /* grow the internal workspace now */
int zk_release_workspace(zk_ctx* ctx);                      /* give the internal workspace back (it regrows on demand) */
int zk_mem_info(zk_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);
int zk_alloc(zk_ctx* ctx, uint64_t bytes, void** d_ptr);
int zk_free(zk_ctx* ctx, void* d_ptr);
int zk_upload(zk_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);
int zk_download(zk_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);
int zk_copy(zk_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes);   /* device to device, async */
/* page-locked host memory: transfers from / to it run at PCIe rate and asynchronously (the ingest path reads files
 * straight into it); zk_upload_async queues the copy on the ctx's stream and returns */
int zk_host_alloc(zk_ctx* ctx, uint64_t bytes, void** ptr);
int zk_host_free(zk_ctx* ctx, void* ptr);
int zk_upload_async(zk_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);

/* Tuning knobs (performance only; results never depend on them). */
#define ZK_TUNE_SORT_VARIANT 1   /* radix-sort geometry index for key arrays, see radix_sort.hip */
#define ZK_TUNE_PAIRS_VARIANT 2  /* ... for (key, payload) pairs */
#define ZK_TUNE_SHORT_SORT 3     /* zk_kmerize: 1 = sort only the top ~log2(n)+3 bits, finish in the mirror stage (default 0) */
#define ZK_TUNE_XCD_GROUP 5    /* radix-sort pipeline: runs of this many consecutive tiles go to one XCD (0 = off, the default; <= 32) */
#define ZK_TUNE_SIDE_DIV 4       /* ... side-list capacity = n / value (default 8); overflow falls back to the full sort */
#define ZK_TUNE_EARLY_COLLAPSE 7 /* zk_kmerize: count the copies of a k-mer before the sort is finished and finish it on (k-mer, count) words:
                                    1 (default) = as early as possible (block hash tables, else tile-local ranking), 3 = tile-local ranking only,
                                    2 = a run-length pass of its own once the copies are neighbours, 0 = off */
#define ZK_TUNE_PACKED_PAIRS 8   /* zk_kmerize / zk_mirror_expand: 1 (default) = (k-mer, count) pairs travel as one 64-bit word when the counts fit the bits above 2K */
#define ZK_TUNE_WIDE_TILES 9     /* radix sort: 1 (default) = array passes use 16 K-key tiles, one 1024-thread workgroup per CU; 0 = 8 K-key tiles, two of 512 */
#define ZK_TUNE_STREAM_PASS 10   /* the first sort pass of zk_kmerize / zk_sort_stream: 1 (default) = static stream ranges, whole 64-byte units written
                                  * out of LDS (stream_pass.hip); 3 = the same, a tile's units leaving in two bursts (measurements; 2 is accepted and
                                  * equals 1); 0 = the look-back pipeline.  Other values are refused */
#define ZK_TUNE_STREAM_RANGES 11 /* ... the number of ranges the stream is cut into, one workgroup each (0 = two per CU, the default; <= 4096) */
#define ZK_TUNE_TAG_WORDS 12     /* zk_kmerize: 1 = the pass before the block dedupe writes 32-bit tags instead of whole keys when the key bits
                                  * below the blocks fit (K <= 25 after two passes); 2 (default) = ... and ranks the keys of a tile by LDS adds
                                  * where the tile lies inside one bucket of the pass before (nobody needs the order inside a block);
                                  * 0 = whole keys */
#define ZK_TUNE_DEDUPE_VARIANT 13 /* the block dedupe of zk_kmerize at <= 32 key bits below the blocks: 0 (default) = dedupe2_kernel, two 512-thread
                                  * workgroups per CU; 1 / 2 = one / two compare-and-swaps in flight per thread instead of four;
                                  * -1 = dedupe_kernel alone (one workgroup per CU, the table of rounds 2 and 3) */
#define ZK_TUNE_DEDUPE_LIMIT 14  /* ... dedupe2_kernel declines blocks of this many keys or more (they are counted by dedupe_kernel afterwards);
                                  * 65536 (default, the most its 16-bit counts allow); tests lower it to reach the second kernel */
#define ZK_TUNE_DEDUPE_BITS 15   /* tests: zk_kmerize takes the block dedupe with this many block bits (a whole number of 9-bit passes, e.g. 18)
                                  * whatever the size of the input, so that small oracle-checked inputs run two passes, tags and tiny blocks; 0 = by size (default) */
#define ZK_TUNE_COMM_CHUNK 6     /* zk_all_to_all_v: bytes per message and round (0 = 256 MiB, the default) */
#define ZK_TUNE_TAG_PASS 17      /* zk_kmerize, the two-pass plan with tags: 1 = pass 0 writes the keys as two arrays (low words, next digits: 6 bytes a key)
                                  * and the second pass is a count / scan / scatter over static segments that writes whole 64-byte units of tags
                                  * (tag_pass.hip); 0 (default) = whole keys and the look-back pipeline.  Same result either way; measured slower
                                  * in all (profiles/r04/tag_pass_ab.json: the second pass gains 5 ms, pass 0 loses 8.5) */
#define ZK_TUNE_KWAY 18          /* zk_merge_n: 1 (default) = inputs of 4 Mi pairs or more are union-summed up to 16 lists at a time, in one pass over the data
                                  * (kway.hip); 2 = every input (tests); 0 = always the tree of 2-way passes */
#define ZK_TUNE_TILE_SORT 19     /* sorts of keys that do not repeat (zk_sort_keys, zk_kmerize on such reads): 1 (default) = LSD passes over the top bits
                                  * only, until blocks of equal top bits are a few dozen keys, then every tile of ~6 K keys sorted to the end in LDS
                                  * (tilesort.hip); 0 = LSD passes over every bit */
#define ZK_TUNE_COMM_SELF_LOOP 16 /* tests: 1 = the piece a rank keeps goes through grouped ncclSend / ncclRecv to itself, in the same rounds as
                                  * the other pieces (instead of a device copy), and zk_allreduce_u64 calls ncclAllReduce with one rank too:
                                  * the RCCL data path of zk_comm_* executed on a box with one GPU; 0 (default) */
int zk_tune(zk_ctx* ctx, int what, int value);

/* Per-launch timing with HIP events recorded on the ctx's own stream (what bench.py's roofline
 * figure is computed from).  Tags name the kernels. */
#define ZK_PROF_HIST_STREAM 1   /* digit histogram straight from the base stream */
#define ZK_PROF_HIST_ARRAY 2
#define ZK_PROF_PASS_STREAM 3   /* radix pass 0: encode in LDS + scatter            (8 B written per key) */
#define ZK_PROF_PASS_KEYS 4     /* radix pass over a key array                      (16 B per key)        */
#define ZK_PROF_PASS_PAIRS 5    /* radix pass over (key, u32) pairs                 (24 B per key)        */
#define ZK_PROF_RLE 6
#define ZK_PROF_UNION 7
#define ZK_PROF_SELECT 8
#define ZK_PROF_MIRROR 9
#define ZK_PROF_INTERSECT 10
#define ZK_PROF_COUNT_HIST 11
#define ZK_PROF_PASS_PACKED 12  /* radix pass of the key kernel over a collapsed list of (k-mer << s | count) words (16 B per word) */
#define ZK_PROF_SAMPLE 13       /* the look before the sort: the few set-aside blocks, sorted and counted (tiny launches) */
#define ZK_PROF_TILE_SORT 14    /* the lower bits of a sort finished tile by tile in LDS (tilesort.hip: 16 B per key, 24 per pair, once) */
int zk_debug_buffer(zk_ctx* ctx, void* d_buf);   /* diagnostic builds (-DZK_STAMPS) only; NULL turns it off */
int zk_profile(zk_ctx* ctx, int enable);   /* clears the records; enable != 0 starts recording */
int zk_profile_read(zk_ctx* ctx, int tag, uint64_t* launches, double* total_ms, uint64_t* algorithmic_bytes);

/* ---- K1/K2: encode ------------------------------------------------------------------------- */

/* reads given as bases[offs[r] .. offs[r+1]) -> base stream (d_stream holds offs[n_reads] + n_reads
 * bytes; read r starts at offs[r] + r and is followed by '\n').  For callers that hold the
 * classic (bases, offsets) layout instead of text lines. */
int zk_pack_reads(zk_ctx* ctx, const uint8_t* d_bases, const uint64_t* d_offs, uint64_t n_reads, uint8_t* d_stream);

/* basics.kmersList(K, seq, both) over every sequence of the stream (library/basics.py:303-347,
 * called from reads.next, library/reads.py:108-117): k-mers in stream order, x then rc(x) per
 * window when both != 0.  acgt (may be NULL) = histogram of the low base of every emitted k-mer
 * (commands/kmerize.py:492-493).  d_out holds cap values; *n_out = values produced. */
int zk_encode(zk_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, int K, int both,
              uint64_t* d_out, uint64_t cap, uint64_t* n_out, uint64_t acgt[4]);

/* basics.sub(seed, p, x) as a filter (library/basics.py:252-259; commands/kmerize.py:494-509):
 * keeps x iff float(murmer(x, seed)) / float(2**61 - 1) < p, compared in doubles. */
int zk_subsample(zk_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint64_t seed, double p,
                 uint64_t* d_out, uint64_t cap, uint64_t* n_out);

/* capture mode, `zot kmerize -C BAITS` (commands/kmerize.py:480-485,510-520): a read is kept iff one
 * of its k-mers is in the sorted bait array (both strands of the bait sequences); reads without a hit
 * are blanked ('N') in d_out, which then feeds zk_kmerize.  A read is a '\n'-terminated piece of the
 * stream.  d_out holds n_bytes. */
int zk_capture_filter(zk_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, int K, const uint64_t* d_baits, uint64_t n_baits,
                      uint8_t* d_out, uint64_t* n_reads, uint64_t* n_kept);

/* basics.can (library/basics.py:231-250; used by `zot vars`): per k-mer, whichever of x and rc(x) has the smaller
 * murmer(., 17) -- x on a tie.  Element-wise, asynchronous; d_out may equal d_kmers. */
int zk_can(zk_ctx* ctx, int K, const uint64_t* d_kmers, uint64_t n, uint64_t* d_out);

/* ---- K3/K4: sort and count ------------------------------------------------------------------ */

/* misc.radix_sort(key_bits, xs) (library/misc.py:400-424): ascending, in place. */
int zk_sort_keys(zk_ctx* ctx, uint64_t* d_keys, uint64_t n, int key_bits);
/* the same carrying a 32-bit payload (stable) */
int zk_sort_pairs(zk_ctx* ctx, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, int key_bits);
/* the run-length half of kmerize.merge (commands/kmerize.py:41-132): distinct values of a sorted
 * array and their multiplicities.  d_uniq may equal d_sorted. */
int zk_rle(zk_ctx* ctx, const uint64_t* d_sorted, uint64_t n, uint64_t* d_uniq, uint32_t* d_counts,
           uint64_t cap, uint64_t* n_unique);
/* KmerAccumulator2.flush on an empty table (commands/kmerize.py:412-424): sort (destroys d_keys)
 * then count. */
int zk_sort_count(zk_ctx* ctx, uint64_t* d_keys, uint64_t n, int key_bits,
                  uint64_t* d_uniq, uint32_t* d_counts, uint64_t cap, uint64_t* n_unique);

/* ---- the fused kmerize batch ------------------------------------------------------------------ */
#define ZK_KMERIZE_CANONICAL 0   /* default: sort one strand, mirror after counting (same result) */
#define ZK_KMERIZE_BOTH 1        /* sort both strands literally, as the reference does */
#define ZK_KMERIZE_SUBSAMPLE 2   /* -D FRAC -S SEED (commands/kmerize.py:469-478,494-509) */
#define ZK_KMERIZE_CANONICAL_ONLY 4   /* stop before the strands are rebuilt: d_kmers / d_counts get the counted CANONICAL list
                                         (c = min(x, rc x) per window, count = windows), n_unique = its length.  The multi-GPU
                                         path exchanges this half-size list and calls zk_mirror_expand on what a rank owns. */

typedef struct {
    uint64_t n_windows;     /* valid windows seen */
    uint64_t n_instances;   /* k-mers the reference would have emitted = 2 * n_windows (kmerize.py:523-525) */
    uint64_t n_unique;      /* entries written to d_kmers / d_counts */
    uint64_t n_canonical;   /* distinct canonical k-mers (0 in ZK_KMERIZE_BOTH mode) */
    uint64_t acgt[4];       /* acgt[x & 3] over every instance, before any filtering (kmerize.py:492-493) */
} zk_kmerize_stats;

/* One in-memory `zot kmerize` (commands/kmerize.py:490-546 with KmerAccumulator2, :370-437) over
 * the base stream: sorted distinct k-mers of BOTH strands and their counts. */
int zk_kmerize(zk_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, int K, int flags, double p, uint64_t seed,
               uint64_t* d_kmers, uint32_t* d_counts, uint64_t cap, zk_kmerize_stats* stats);

/* The second half of zk_kmerize for a counted canonical list (ascending c, counts): sorted distinct k-mers of BOTH strands
 * -- (c, n) and (rc c, n) for every entry, a palindrome (c == rc c) counted n + n, as two emissions per window give
 * (commands/kmerize.py:490; library/reads.py:113-114).  zk_kmerize(flags) == zk_mirror_expand(zk_kmerize(flags |
 * ZK_KMERIZE_CANONICAL_ONLY)). */
int zk_mirror_expand(zk_ctx* ctx, const uint64_t* d_canon_kmers, const uint32_t* d_canon_counts, uint64_t n, int K,
                     uint64_t* d_kmers, uint32_t* d_counts, uint64_t cap, uint64_t* n_out);

/* hist[c] += 1 per distinct k-mer (commands/kmerize.py:543-545; merge.py:88-92), as ascending
 * (value, frequency) pairs in HOST arrays of cap_bins entries.  count_bits is 32 or 64. */
int zk_hist(zk_ctx* ctx, const void* d_counts, int count_bits, uint64_t n,
            uint64_t* vals, uint64_t* freq, uint64_t cap_bins, uint64_t* n_bins);

/* uint32 counts -> uint64 counts (kmerize keeps array('I'), merge works on Python ints) */
int zk_widen_counts(zk_ctx* ctx, const uint32_t* d_in, uint64_t* d_out, uint64_t n);

/* ---- K5/K6: union with summed counts ---------------------------------------------------------- */

/* merge.merge (commands/merge.py:26-86) and the merge half of kmerize.merge: two sorted-unique
 * (k-mer, count) lists -> one; equal k-mers add.  count_bits (32 or 64) is the element type of all
 * three count arrays.  acgt_w (may be NULL): acgt_w[x & 3] += count (merge.py:159). */
int zk_union_sum(zk_ctx* ctx, const uint64_t* d_xk, const void* d_xc, uint64_t nx,
                 const uint64_t* d_yk, const void* d_yc, uint64_t ny,
                 uint64_t* d_ok, void* d_oc, int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]);

/* mergeNinto (commands/merge.py:127-163; twin commands/kmerize.py:269-304): k sorted-unique lists
 * -> one.  d_keys / d_counts / ns are HOST arrays of k device pointers / sizes; count_bits (32 or 64)
 * is the element type of every count array. */
int zk_merge_n(zk_ctx* ctx, int k, const uint64_t* const* d_keys, const void* const* d_counts, const uint64_t* ns,
               uint64_t* d_ok, void* d_oc, int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]);

/* ---- K8/K9: dist ------------------------------------------------------------------------------- */

/* Measure.prep, set mode (commands/dist.py:43-49): y = x >> shift, adjacent duplicates dropped. */
int zk_project_dedupe(zk_ctx* ctx, const uint64_t* d_kmers, uint64_t n, int shift,
                      uint64_t* d_out, uint64_t cap, uint64_t* n_out);
/* dist.split (library/dist.py:241-265): abc = (|X & Y|, |X \ Y|, |Y \ X|) of two sorted unique arrays. */
int zk_split(zk_ctx* ctx, const uint64_t* d_x, uint64_t nx, const uint64_t* d_y, uint64_t ny, uint64_t abc[3]);

/* positions[q] = how many elements of the sorted device array are < queries[q] (HOST arrays of m
 * entries): the cut points of a value-range partition for the multi-GPU exchange. */
int zk_lower_bound(zk_ctx* ctx, const uint64_t* d_sorted, uint64_t n, const uint64_t* queries, uint32_t m, uint64_t* positions);

/* ---- next-row commands on the same kernels ---------------------------------------- */

/* project.project2 (commands/project.py:29-40): the (k-mer, count) entries of a set whose k-mer is in the
 * sorted reference set; 64-bit counts. */
int zk_project(zk_ctx* ctx, const uint64_t* d_ref, uint64_t n_ref, const uint64_t* d_kmers, const uint64_t* d_counts, uint64_t n,
               uint64_t* d_ok, uint64_t* d_oc, uint64_t cap, uint64_t* n_out);
/* sample.sampleD (commands/sample.py:27-34): keep iff float(murmer(x, seed) & (2^40 - 1)) / float(2^40 - 1) < p. */
int zk_sample(zk_ctx* ctx, const uint64_t* d_kmers, const uint64_t* d_counts, uint64_t n, uint64_t seed, double p,
              uint64_t* d_ok, uint64_t* d_oc, uint64_t cap, uint64_t* n_out);

/* ---- multi-GPU: partitioning and the RCCL seam ------------------------------------- */

/* Hash-range owner: owner(x) = floor(murmer(x, seed) * world / 2^64) (basics.murmer, library/basics.py:191-229).
 * Splits a sorted (k-mer, count) table into `world` pieces, each still sorted (a stable partition); piece o is
 * d_ok[offsets[o] .. offsets[o+1]).  d_counts / d_oc may be NULL (keys only); offsets is a HOST array of world + 1
 * entries; world <= 32.  The value-range owner needs no kernel: a sorted table is already partitioned, the cut
 * points come from zk_lower_bound. */
int zk_hash_partition(zk_ctx* ctx, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n, int world, uint64_t seed,
                      uint64_t* d_ok, void* d_oc, uint64_t* offsets);

/* One RCCL communicator per context, one process per GPU.  The reference is single-process (no counterpart);
 * these are the entry points a host that is not PyTorch binds to drive the 8-GPU path.  Rank 0 makes the id and
 * the host carries its 128 bytes to the other ranks (file, MPI, torch.distributed store ...). */
#define ZK_COMM_ID_BYTES 128
#define ZK_REDUCE_SUM 0          /* modulo 2^64 */
#define ZK_REDUCE_MAX 1
int zk_comm_unique_id(uint8_t id[ZK_COMM_ID_BYTES]);
int zk_comm_init(zk_ctx* ctx, int world, int rank, const uint8_t id[ZK_COMM_ID_BYTES]);
int zk_comm_destroy(zk_ctx* ctx);
int zk_comm_info(zk_ctx* ctx, int* world, int* rank);          /* 1, 0 without a communicator */
/* The exchange step: elements [send_off[r], send_off[r] + send_cnt[r]) of d_send go to rank r, which receives them
 * at recv_off[me] of its d_recv (recv_cnt[r] on this rank must equal send_cnt[me] on rank r); offsets and counts are
 * HOST arrays of `world` entries in units of elem_bytes.  Grouped ncclSend / ncclRecv, one hop per peer on the xGMI
 * mesh, straight between the callers' arrays, <= 256 MiB per message and round; asynchronous on the ctx's stream. */
int zk_all_to_all_v(zk_ctx* ctx, const void* d_send, const uint64_t* send_off, const uint64_t* send_cnt, void* d_recv,
                    const uint64_t* recv_off, const uint64_t* recv_cnt, int elem_bytes);
/* The messages zk_all_to_all_v issues, as data (host only; no GPU, no RCCL needed): rank `rank` of `world`, same arguments; chunk_bytes 0 =
 * 256 MiB; self_loop as ZK_TUNE_COMM_SELF_LOOP.  ops[0 .. min(*n_ops, cap)) in issue order: recv 0 = a send of bytes [offset, offset + bytes)
 * of d_send to `peer`, 1 = a receive into those bytes of d_recv from `peer`; all the ops of one `round` form one ncclGroup.  For tests: what
 * a box with one GPU cannot execute -- the pairing of the ranks and the offsets of their pieces -- is checked on the CPU. */
typedef struct { int32_t recv; int32_t peer; uint64_t round; uint64_t offset; uint64_t bytes; } zk_comm_op;
int zk_comm_plan(int world, int rank, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off, const uint64_t* recv_cnt,
                 int elem_bytes, uint64_t chunk_bytes, int self_loop, zk_comm_op* ops, uint64_t cap, uint64_t* n_ops);
/* in-place all-reduce of n HOST values (dist's (a, b, c), checksums, the splitter histogram); synchronises */
int zk_allreduce_u64(zk_ctx* ctx, uint64_t* vals, uint64_t n, int op);

/* ---- K10: trim ---------------------------------------------------------------------------------- */

/* trim.trim (commands/trim.py:54-62): keep (x, f) iff f >= lo and (hi == 0 or f <= hi). */
int zk_trim(zk_ctx* ctx, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n,
            uint64_t lo, uint64_t hi, uint64_t* d_ok, void* d_oc, uint64_t cap, uint64_t* n_out);

/* ---- host-side format code (CPU; no ctx, no GPU): next-row f1/f2 of SURVEY.md section 8 ------------ */

/* codec64.encode (library/codec64.py:42-120) over HOST arrays; delta != 0 first replaces ascending
 * k-mers by their differences (files.delta, library/files.py:85-98).  ZK_ERANGE when a value or delta
 * is >= 2^60 (the reference raises there).  words needs at most n entries. */
int zk_codec64_encode(const uint64_t* vals, uint64_t n, int delta, uint64_t* words, uint64_t cap, uint64_t* n_words);
/* sum of the tags = number of values in a word stream */
int zk_codec64_count(const uint64_t* words, uint64_t nw, uint64_t* n_values);
/* codec64.decode (library/codec64.py:122-151); delta != 0 also undoes the delta transform
 * (files.undelta, library/files.py:100-110) */
int zk_codec64_decode(const uint64_t* words, uint64_t nw, int delta, uint64_t* out, uint64_t cap, uint64_t* n_out);

/* The same codec on the device (K11 / K12): values and words are DEVICE arrays; streams are byte-exact
 * with the host functions above and with the reference.  encode: d_words needs at most n entries;
 * decode: the exact value count comes back in *n_out, also with ZK_ENOSPC when it exceeds cap (so a first
 * call with cap = 0 sizes the output). */
int zk_codec64_encode_dev(zk_ctx* ctx, const uint64_t* d_vals, uint64_t n, int delta, uint64_t* d_words, uint64_t cap, uint64_t* n_words);
int zk_codec64_decode_dev(zk_ctx* ctx, const uint64_t* d_words, uint64_t nw, int delta, uint64_t* d_out, uint64_t cap, uint64_t* n_out);
/* zk_codec64_encode_dev (delta = 0) of 32-bit values: the counts of `zot kmerize` (array('I'), commands/kmerize.py:370-437) go into
 * the 'counts' member (files.writeKmersAndCounts2, library/files.py:209-217) without being widened to 64 bits first; same words. */
int zk_codec64_encode_u32_dev(zk_ctx* ctx, const uint32_t* d_vals, uint64_t n, uint64_t* d_words, uint64_t cap, uint64_t* n_words);

/* files.undelta (library/files.py:100-110) of a PIECE of a delta stream: in-place inclusive prefix sum of the decoded
 * deltas, continued from `base` (the last k-mer before the piece; 0 for the first piece).  With zk_add_u64 (v[i] += x,
 * asynchronous) this lets every rank of a multi-GPU `zot dist` decode its own contiguous range of the words of a
 * 'kmers' member: scan from 0, exchange the pieces' totals, add the base. */
int zk_undelta(zk_ctx* ctx, uint64_t* d_vals, uint64_t n, uint64_t base);
int zk_add_u64(zk_ctx* ctx, uint64_t* d_vals, uint64_t n, uint64_t x);

/* file.readFastq / file.readFasta (library/file.py:19-52) as chunked text -> base stream converters.
 * state: 4 words, zero before the first chunk, state[1] = records seen.  Feed text; *consumed says how
 * much was used (present the rest again in front of the next chunk); final != 0 on the last chunk.
 * *out_len is read (append position) and updated. */
int zk_parse_fastq(const char* buf, uint64_t len, int final, uint64_t state[4], uint8_t* out, uint64_t out_cap,
                   uint64_t* out_len, uint64_t* consumed);
int zk_parse_fasta(const char* buf, uint64_t len, int final, uint64_t state[4], uint8_t* out, uint64_t out_cap,
                   uint64_t* out_len, uint64_t* consumed);

/* FASTQ text on the device -> base stream of the same length (file.readFastq, library/file.py:38-52):
 * every byte that is not on the second line of a group of four becomes '\n'.  line_phase = number of
 * complete lines before this text, mod 4 (feed whole lines); *n_newlines = '\n' bytes in the text.
 * d_stream may not alias d_text; both 16-byte aligned. */
int zk_fastq_mask(zk_ctx* ctx, const uint8_t* d_text, uint64_t n, uint32_t line_phase, uint8_t* d_stream, uint64_t* n_newlines);

/* ---- ingest: file bytes <-> device memory (next-row f2 of SURVEY.md section 8) --------------------------------------
 * Replaces file.readFastq / openFile + `gunzip -c` (library/file.py:38-52,79-123) feeding reads.next (library/reads.py:86-98):
 * the text is parsed on the device (zk_fastq_mask), the host only moves bytes.  A zk_source reads a file -- plain, or gzip
 * (detected by its magic; multi-member files included) -- AHEAD of the device: a background thread fills page-locked
 * buffers (parallel pread, or zlib inflate) and queues them as asynchronous H2D copies on a copy stream of its own, so
 * reading / inflating, PCIe and the kernels working on the previous batch overlap.  zk_source_start names the device
 * buffer of the next batch and returns at once; zk_source_finish waits and says how many bytes arrived (fewer than
 * asked only at the end of the input).  Batches are cut at line ends with zk_last_newline (position just after the last
 * newline of d_text[0, n), 0 if none) and the tail is carried to the front of the next buffer by the caller (zk_copy). */
typedef struct zk_source zk_source;
zk_source* zk_source_open(zk_ctx* ctx, const char* path, int threads);      /* NULL on failure: zk_last_error(ctx) */
int zk_source_is_gzip(zk_source* src);
int zk_source_start(zk_source* src, uint8_t* d_dst, uint64_t cap);
int zk_source_finish(zk_source* src, uint64_t* n_bytes, int* eof);
void zk_source_close(zk_source* src);
int zk_last_newline(zk_ctx* ctx, const uint8_t* d_text, uint64_t n, uint64_t* cut);
/* a member of a k-mer set between device memory and a region of an open file (files.writeWords / readWords,
 * library/files.py:54-83): D2H / H2D through the page-locked ring, pwrite / pread in `threads` parallel slices */
int zk_device_to_file(zk_ctx* ctx, const void* d_src, uint64_t bytes, int fd, uint64_t file_offset, int threads);
int zk_file_to_device(zk_ctx* ctx, int fd, uint64_t file_offset, uint64_t bytes, void* d_dst, int threads);

/* ---- synthetic input (bench / tests) ------------------------------------- */

/* Reads first .. first+count-1 of the counter-based generator (zotmer_amd/synth.py) as a base
 * stream of count*(L+1) bytes.  genome == 0: uniform bases.  Thresholds are fractions of 2^32. */
int zk_synth_reads(zk_ctx* ctx, uint64_t seed, uint64_t first, uint64_t count, int L, uint64_t genome,
                   uint32_t sub_thr, uint32_t n_thr, uint8_t* d_stream);

/* sum of x and of murmer(x, 0) * count over a counted set, mod 2^64: order-free checksums used by
 * the full-size parity tests.  d_counts may be NULL (every count 1). */
int zk_checksum(zk_ctx* ctx, const uint64_t* d_kmers, const uint32_t* d_counts, uint64_t n, uint64_t sums[3]);

/* The sorted-set format stores strictly ascending k-mers (library/files.py:54-110 delta-codes them; every consumer merges on
 * that order, commands/merge.py:26-86).  *first_bad = the first index i >= 1 with d_kmers[i] <= d_kmers[i - 1], or n if the
 * array is strictly ascending: the order check of the full-size parity tests (the checksums are order-free). */
int zk_first_descent(zk_ctx* ctx, const uint64_t* d_kmers, uint64_t n, uint64_t* first_bad);

/* zk_checksum over 32- or 64-bit counts (`zot merge` works on 64-bit counts) */
int zk_checksum_counts(zk_ctx* ctx, const uint64_t* d_kmers, const void* d_counts, int count_bits, uint64_t n, uint64_t sums[3]);

/* Synthetic k-mer sets (BASELINE configs 3 and 4; zotmer_amd/synth.py set_keys / set_counts is the specification).
 * d_out[i] = rnd(seed, 7, (mul * (first + i) + add) % mod) & (2^key_bits - 1): element first + i of an affine walk through
 * a pool of `mod` random keys (unsorted; sort and dedupe with zk_sort_count).  zk_synth_counts: geometric counts
 * (mean 8) as a function of (seed, key). */
int zk_synth_keys(zk_ctx* ctx, uint64_t seed, uint64_t first, uint64_t count, int key_bits, uint64_t mul, uint64_t add, uint64_t mod,
                  uint64_t* d_out);
int zk_synth_counts(zk_ctx* ctx, uint64_t seed, const uint64_t* d_keys, uint64_t n, uint64_t* d_counts);

/* the same three sums over every k-mer instance (x and rc(x) of each valid window) of a base
 * stream, computed straight from the stream without sorting anything; sums[3..6] = acgt[x & 3]
 * over the same instances (commands/kmerize.py:492-493) */
int zk_stream_checksum(zk_ctx* ctx, const uint8_t* d_stream, uint64_t n_bytes, int K, uint64_t sums[7]);

/* internal key-source selectors (shared with the kernels) */
#define ZK_KEYS_FORWARD 0
#define ZK_KEYS_BOTH 1
#define ZK_KEYS_CANONICAL 2

#ifdef __cplusplus
}
#endif
#endif /* ZOTK_H */
