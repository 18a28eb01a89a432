/*
 * zk_oracle.c -- CPU restatement of the zotmer kmerize / merge / dist / trim hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under zotmer_amd/ may import, link or call this
 * file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker (or as the reported single-core CPU baseline), never as the
 * thing that is shipped or measured as the product.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against golden vectors captured from the reference itself (tests/golden/make_golden.py
 * imports /root/reference's unmodified arithmetic modules under Python 3 and drives the
 * reference commands through a container-local 2to3 copy; see tests/golden/README.md).
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 * It is a restatement in plain C of what those lines compute, not a transliteration.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/libzkoracle.so oracle/zk_oracle.c   (see oracle/Makefile)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;

/* ------------------------------------------------------------------------------------------
 * bit tricks: zotmer/library/bits.py
 * ---------------------------------------------------------------------------------------- */

/* bits.rev (zotmer/library/bits.py:22-31): reverse the order of the 32 bit-pairs of a word. */
u64 zo_rev(u64 x) {
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    return x;
}

/* bits.popcnt (zotmer/library/bits.py:33-43). */
int zo_popcnt(u64 x) {
    int n = 0;
    while (x) { x &= x - 1; n++; }
    return n;
}

/* bits.ffs (zotmer/library/bits.py:64-98): despite the name, index of the MOST significant
 * set bit; 0 for x == 0 (the table entry _ffsBits[0] is 0). */
int zo_ffs(u64 x) {
    int r = 0;
    while (x > 1) { x >>= 1; r++; }
    return r;
}

/* ------------------------------------------------------------------------------------------
 * k-mer primitives: zotmer/library/basics.py
 * ---------------------------------------------------------------------------------------- */

/* basics._nuc (zotmer/library/basics.py:42-46): AaCcGgTtUu -> 0,1,2,3 ; anything else invalid. */
static int nuc_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return -1;
    }
}

/* basics.kmer (zotmer/library/basics.py:48-59): returns 0 and sets *ok=0 on an invalid base. */
u64 zo_kmer(const char* seq, int len, int* ok) {
    u64 r = 0;
    *ok = 1;
    for (int i = 0; i < len; i++) {
        int b = nuc_code((unsigned char)seq[i]);
        if (b < 0) { *ok = 0; return 0; }
        r = (r << 2) | (u64)b;
    }
    return r;
}

/* basics.render (zotmer/library/basics.py:61-67): out must hold k+1 bytes. */
void zo_render(int k, u64 x, char* out) {
    for (int i = k - 1; i >= 0; i--) { out[i] = "ACGT"[x & 3]; x >>= 2; }
    out[k] = 0;
}

/* basics.rc (zotmer/library/basics.py:115-121): rev(~x) >> (64 - 2k). */
u64 zo_rc(int k, u64 x) {
    return zo_rev(~x) >> (64 - 2 * k);
}

/* basics.ham (zotmer/library/basics.py:123-133). */
int zo_ham(u64 x, u64 y) {
    u64 z = x ^ y;
    return zo_popcnt((z | (z >> 1)) & 0x5555555555555555ULL);
}

/* basics.lcp (zotmer/library/basics.py:160-170). */
int zo_lcp(int k, u64 x, u64 y) {
    u64 z = x ^ y;
    if (z == 0) return k;
    return k - (1 + zo_ffs(z) / 2);
}

/* basics.fnv (zotmer/library/basics.py:172-189): FNV-1a style over seed bytes then k-mer
 * bytes, truncated to 61 bits after every multiply. */
u64 zo_fnv(u64 x, u64 s) {
    const u64 M61 = 0x1FFFFFFFFFFFFFFFULL;
    u64 h = 0xcbf29ce484222325ULL;
    for (int i = 0; i < 8; i++) { h ^= (s & 0xff); h = (h * 0x100000001b3ULL) & M61; s >>= 8; }
    for (int i = 0; i < 8; i++) { h ^= (x & 0xff); h = (h * 0x100000001b3ULL) & M61; x >>= 8; }
    return h & M61;
}

/* basics.murmer (zotmer/library/basics.py:191-229): one MurmurHash3-x64 block of (x, seed),
 * full 64-bit result (the docstring's "61 bits" is not applied). */
u64 zo_murmer(u64 x, u64 s) {
    u64 k = x * 0x87c37b91114253d5ULL;
    k = (k << 31) | (k >> 33);
    k *= 0x4cf5ad432745937fULL;
    u64 h = s ^ k;
    h = (h << 27) | (h >> 37);
    h = h * 5 + 0x52dce729ULL;
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return h;
}

/* basics.can (zotmer/library/basics.py:231-250). */
u64 zo_can(int k, u64 x) {
    u64 xb = zo_rc(k, x);
    return (zo_murmer(x, 17) <= zo_murmer(xb, 17)) ? x : xb;
}

/* basics.sub (zotmer/library/basics.py:252-259): float(h)/float(2^61-1) < p, as doubles. */
int zo_sub(u64 s, double p, u64 x) {
    double u = (double)zo_murmer(x, s) / (double)0x1FFFFFFFFFFFFFFFULL;
    return u < p;
}

/* basics.kmersList (zotmer/library/basics.py:303-347): sliding window over one sequence;
 * a byte outside AaCcGgTtUu restarts the window after it; with both != 0 every window
 * contributes x then its reverse complement xb (kept incrementally, :337-338).
 * Returns the number of values written (never more than cap; the needed size is
 * (both ? 2 : 1) * max(0, len - k + 1)). */
u64 zo_kmers_list(int k, const char* seq, u64 len, int both, u64* out, u64 cap) {
    const u64 msk = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const int s = 2 * (k - 1);
    u64 x = 0, xb = 0, n = 0;
    int j = 0; /* number of valid bases currently in the window */
    for (u64 p = 0; p < len; p++) {
        int b = nuc_code((unsigned char)seq[p]);
        if (b < 0) { j = 0; x = 0; xb = 0; continue; }
        x = ((x << 2) | (u64)b) & msk;
        xb = (xb >> 2) | ((u64)(3 - b) << s);
        if (j < k) j++;
        if (j == k) {
            if (n < cap) out[n] = x;
            n++;
            if (both) { if (n < cap) out[n] = xb; n++; }
        }
    }
    return n < cap ? n : cap;
}

/* ------------------------------------------------------------------------------------------
 * sort + run-length count: zotmer/library/misc.py, zotmer/commands/kmerize.py
 * ---------------------------------------------------------------------------------------- */

static int cmp_u64(const void* a, const void* b) {
    u64 x = *(const u64*)a, y = *(const u64*)b;
    return (x > y) - (x < y);
}

/* misc.radix_sort (zotmer/library/misc.py:400-424): MSD split on 8-bit digits of the `bits`
 * significant bits for at most two levels (and only while more than 16384 items remain),
 * then a comparison sort of each bucket.  Net effect: xs ascending. */
static void radix_rec(int bits, int d, u64* xs, u64 n, u64* tmp) {
    if (d >= 2 || (d + 1) * 8 >= bits || n <= 16384) { qsort(xs, n, sizeof(u64), cmp_u64); return; }
    int s = bits - (d + 1) * 8;
    u64 cnt[257];
    memset(cnt, 0, sizeof cnt);
    for (u64 i = 0; i < n; i++) cnt[((xs[i] >> s) & 255) + 1]++;
    for (int p = 0; p < 256; p++) cnt[p + 1] += cnt[p];
    u64 pos[256];
    for (int p = 0; p < 256; p++) pos[p] = cnt[p];
    for (u64 i = 0; i < n; i++) tmp[pos[(xs[i] >> s) & 255]++] = xs[i];
    memcpy(xs, tmp, n * sizeof(u64));
    for (int p = 0; p < 256; p++) radix_rec(bits, d + 1, xs + cnt[p], cnt[p + 1] - cnt[p], tmp + cnt[p]);
}

int zo_radix_sort(int bits, u64* xs, u64 n) {
    if (n <= 16384) { qsort(xs, n, sizeof(u64), cmp_u64); return 0; }
    u64* tmp = (u64*)malloc(n * sizeof(u64));
    if (!tmp) return -1;
    radix_rec(bits, 0, xs, n, tmp);
    free(tmp);
    return 0;
}

/* kmerize.merge (zotmer/commands/kmerize.py:41-132): run-length count the sorted raw list ys
 * while 2-way merging it into the sorted-unique (xs, cs); equal keys add.  zs/ss must hold
 * nx + ny entries.  Returns the merged length.  Counts are 32-bit as in the reference's
 * array('I') (:373-374,418-419); *overflow is set if one would not fit. */
u64 zo_rle_merge(const u64* xs, const u32* cs, u64 nx, const u64* ys, u64 ny,
                 u64* zs, u32* ss, int* overflow) {
    u64 i = 0, j = 0, n = 0;
    if (overflow) *overflow = 0;
    while (i < nx || j < ny) {
        u64 key; u64 c = 0;
        if (j >= ny || (i < nx && xs[i] < ys[j])) { key = xs[i]; c = cs[i]; i++; }
        else {
            key = ys[j];
            while (j < ny && ys[j] == key) { c++; j++; }
            if (i < nx && xs[i] == key) { c += cs[i]; i++; }
        }
        if (c > 0xFFFFFFFFULL && overflow) *overflow = 1;
        zs[n] = key; ss[n] = (u32)c; n++;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * zot kmerize in memory: zotmer/commands/kmerize.py:450-562 with KmerAccumulator2 (:370-437)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int K;
    u64* idxX; u32* idxC; u64 nIdx;      /* sorted-unique table        (:373-374) */
    u64* buf; u64 nBuf, capBuf;          /* unsorted pending instances (:375)     */
    u64 flushAt;                         /* reference: 128 Mi (:398,409)          */
    int overflow;
} zo_acc;

static int acc_flush(zo_acc* a) {      /* KmerAccumulator2.flush (:412-424) */
    if (a->nBuf == 0) return 0;
    if (zo_radix_sort(2 * a->K, a->buf, a->nBuf)) return -1;
    u64* zs = (u64*)malloc((a->nIdx + a->nBuf) * sizeof(u64));
    u32* ss = (u32*)malloc((a->nIdx + a->nBuf) * sizeof(u32));
    if (!zs || !ss) { free(zs); free(ss); return -1; }
    int ov = 0;
    u64 n = zo_rle_merge(a->idxX, a->idxC, a->nIdx, a->buf, a->nBuf, zs, ss, &ov);
    if (ov) a->overflow = 1;
    free(a->idxX); free(a->idxC);
    a->idxX = zs; a->idxC = ss; a->nIdx = n; a->nBuf = 0;
    return 0;
}

static int acc_add(zo_acc* a, const u64* xs, u64 n) {   /* addList (:401-410) */
    if (a->nBuf + n > a->capBuf) {
        u64 cap = a->capBuf ? a->capBuf : 1024;
        while (cap < a->nBuf + n) cap *= 2;
        u64* nb = (u64*)realloc(a->buf, cap * sizeof(u64));
        if (!nb) return -1;
        a->buf = nb; a->capBuf = cap;
    }
    memcpy(a->buf + a->nBuf, xs, n * sizeof(u64));
    a->nBuf += n;
    if (a->nBuf > a->nIdx && a->nBuf > a->flushAt) return acc_flush(a);
    return 0;
}

/* Result block returned by zo_kmerize; arrays are malloc'ed, release with zo_kmerize_free. */
typedef struct {
    u64* kmers; u32* counts; u64 n_unique;
    u64 acgt[4];         /* acgt[x&3] over every emitted instance BEFORE filtering (:492-493) */
    u64 n_reads;         /* records seen, including ones that yield nothing (:527)            */
    u64 n_kept;          /* instances that reached the accumulator                            */
    int overflow;
} zo_kmerize_result;

/* One in-memory `zot kmerize` over reads given as bases[offs[r] .. offs[r+1]).
 *   mode 0: plain                       (:521-525)
 *   mode 1: -D subsample, sub(seed,p,x) per instance            (:494-509)
 *   mode 2: -C capture, baits = sorted both-strand bait k-mers; a read contributes all its
 *           k-mers iff any one of them is a bait                (:510-520)
 * flush_at = 0 selects the reference's 128 Mi threshold; tests pass small values to exercise
 * the flush/merge path (the final arrays do not depend on it). */
int zo_kmerize(int K, const char* bases, const u64* offs, u64 n_reads, int mode,
               double p, u64 seed, const u64* baits, u64 n_baits, u64 flush_at,
               zo_kmerize_result* res) {
    zo_acc a;
    memset(&a, 0, sizeof a);
    memset(res, 0, sizeof *res);
    a.K = K;
    a.flushAt = flush_at ? flush_at : (128ULL * 1024 * 1024);
    u64 capx = 1024;
    u64* xs = (u64*)malloc(capx * sizeof(u64));
    if (!xs) return -1;
    int rc = 0;
    for (u64 r = 0; r < n_reads && rc == 0; r++) {
        u64 len = offs[r + 1] - offs[r];
        u64 need = (len >= (u64)K) ? 2 * (len - K + 1) : 0;
        if (need > capx) { while (capx < need) capx *= 2; free(xs); xs = (u64*)malloc(capx * sizeof(u64)); if (!xs) return -1; }
        u64 n = zo_kmers_list(K, bases + offs[r], len, 1, xs, capx);
        for (u64 i = 0; i < n; i++) res->acgt[xs[i] & 3]++;
        if (mode == 1) {
            u64 m = 0;
            for (u64 i = 0; i < n; i++) if (zo_sub(seed, p, xs[i])) xs[m++] = xs[i];
            n = m;
        } else if (mode == 2) {
            int found = 0;
            for (u64 i = 0; i < n && !found; i++) {
                u64 lo = 0, hi = n_baits;
                while (lo < hi) { u64 mid = (lo + hi) / 2; if (baits[mid] < xs[i]) lo = mid + 1; else hi = mid; }
                found = (lo < n_baits && baits[lo] == xs[i]);
            }
            if (!found) n = 0;
        }
        res->n_kept += n;
        if (n) rc = acc_add(&a, xs, n);
        res->n_reads++;
    }
    if (rc == 0) rc = acc_flush(&a);          /* kmersOnly()/countsOnly() flush (:431-437) */
    free(xs); free(a.buf);
    if (rc) { free(a.idxX); free(a.idxC); return rc; }
    res->kmers = a.idxX; res->counts = a.idxC; res->n_unique = a.nIdx; res->overflow = a.overflow;
    return 0;
}

void zo_kmerize_free(zo_kmerize_result* res) {
    free(res->kmers); free(res->counts);
    res->kmers = 0; res->counts = 0; res->n_unique = 0;
}

/* hist (zotmer/commands/kmerize.py:543-545; merge.py:88-92): h[c] += 1 per distinct k-mer.
 * Written as ascending (value, freq) pairs; returns the number of pairs (<= n). */
u64 zo_hist(const u64* counts, u64 n, u64* vals, u64* freq) {
    if (n == 0) return 0;
    u64* t = (u64*)malloc(n * sizeof(u64));
    memcpy(t, counts, n * sizeof(u64));
    qsort(t, n, sizeof(u64), cmp_u64);
    u64 m = 0;
    for (u64 i = 0; i < n;) {
        u64 j = i;
        while (j < n && t[j] == t[i]) j++;
        vals[m] = t[i]; freq[m] = j - i; m++;
        i = j;
    }
    free(t);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * zot merge: zotmer/commands/merge.py
 * ---------------------------------------------------------------------------------------- */

/* merge.merge (zotmer/commands/merge.py:26-86): streaming 2-way union of sorted-unique
 * (key,count) lists; equal keys add.  zs/zc hold nx + ny.  Returns the output length. */
u64 zo_union_sum(const u64* xs, const u64* xc, u64 nx, const u64* ys, const u64* yc, u64 ny,
                 u64* zs, u64* zc) {
    u64 i = 0, j = 0, n = 0;
    while (i < nx && j < ny) {
        if (xs[i] < ys[j]) { zs[n] = xs[i]; zc[n] = xc[i]; i++; }
        else if (xs[i] > ys[j]) { zs[n] = ys[j]; zc[n] = yc[j]; j++; }
        else { zs[n] = xs[i]; zc[n] = xc[i] + yc[j]; i++; j++; }
        n++;
    }
    for (; i < nx; i++, n++) { zs[n] = xs[i]; zc[n] = xc[i]; }
    for (; j < ny; j++, n++) { zs[n] = ys[j]; zc[n] = yc[j]; }
    return n;
}

typedef struct { u64 k, c; } zo_pair;
static int cmp_pair(const void* a, const void* b) {
    u64 x = ((const zo_pair*)a)->k, y = ((const zo_pair*)b)->k;
    return (x > y) - (x < y);
}

/* mergeNinto (zotmer/commands/merge.py:127-163, twin zotmer/commands/kmerize.py:269-304):
 * k-way union with summed counts, taken one top-12-bit radix block at a time
 * (_kmerRadixBlockStream, merge.py:94-125): the items of every stream whose key falls in
 * the block are pooled, equal keys summed (the reference's dict), sorted, emitted; acgt_w
 * accumulates acgt[x&3] += c per emitted pair (merge.py:159).  zs/zc hold sum(ns). */
u64 zo_merge_n(int K, int k, const u64* const* xs, const u64* const* xc, const u64* ns,
               u64* zs, u64* zc, u64 acgt_w[4]) {
    int S = 2 * K - 12;
    if (S < 0) S = 0;
    u64* pos = (u64*)calloc(k, sizeof(u64));
    u64 total = 0, n = 0;
    for (int s = 0; s < k; s++) total += ns[s];
    zo_pair* pool = (zo_pair*)malloc((total ? total : 1) * sizeof(zo_pair));
    if (acgt_w) memset(acgt_w, 0, 4 * sizeof(u64));
    u64 done = 0;
    for (u64 radix = 0; done < total; radix++) {
        u64 m = 0;
        for (int s = 0; s < k; s++) {
            while (pos[s] < ns[s] && (xs[s][pos[s]] >> S) == radix) {
                pool[m].k = xs[s][pos[s]]; pool[m].c = xc[s][pos[s]]; m++; pos[s]++; done++;
            }
        }
        qsort(pool, m, sizeof(zo_pair), cmp_pair);
        for (u64 i = 0; i < m;) {
            u64 c = 0, j = i;
            while (j < m && pool[j].k == pool[i].k) { c += pool[j].c; j++; }
            zs[n] = pool[i].k; zc[n] = c; n++;
            if (acgt_w) acgt_w[pool[i].k & 3] += c;
            i = j;
        }
    }
    free(pool); free(pos);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * zot dist: zotmer/commands/dist.py, zotmer/library/dist.py
 * ---------------------------------------------------------------------------------------- */

/* Measure.prep, set mode (zotmer/commands/dist.py:43-49): y = x >> shift, keep if it differs
 * from the previously kept value.  shift = 2*(fileK - K). */
u64 zo_project_dedupe(const u64* xs, u64 n, int shift, u64* out) {
    u64 m = 0;
    for (u64 i = 0; i < n; i++) {
        u64 y = xs[i] >> shift;
        if (m == 0 || out[m - 1] != y) out[m++] = y;
    }
    return m;
}

/* dist.split (zotmer/library/dist.py:241-265): abc = (|X&Y|, |X\Y|, |Y\X|). */
void zo_split(const u64* xs, u64 nx, const u64* ys, u64 ny, u64 abc[3]) {
    u64 i = 0, j = 0, both = 0, dx = 0, dy = 0;
    while (i < nx && j < ny) {
        if (xs[i] < ys[j]) { dx++; i++; }
        else if (xs[i] > ys[j]) { dy++; j++; }
        else { both++; i++; j++; }
    }
    abc[0] = both; abc[1] = dx + (nx - i); abc[2] = dy + (ny - j);
}

/* ------------------------------------------------------------------------------------------
 * next-row commands on the same sets: zot project, zot sample
 * ---------------------------------------------------------------------------------------- */

/* project.project2 (zotmer/commands/project.py:29-40): the entries (y, c) of the input whose k-mer is
 * in the sorted reference xs. */
u64 zo_project(const u64* xs, u64 nx, const u64* ys, const u64* yc, u64 ny, u64* ok, u64* oc) {
    u64 i = 0, m = 0;
    for (u64 j = 0; j < ny; j++) {
        while (i < nx && xs[i] < ys[j]) i++;
        if (i == nx) break;
        if (xs[i] == ys[j]) { ok[m] = ys[j]; oc[m] = yc[j]; m++; i++; }
    }
    return m;
}

/* sample.sampleD (zotmer/commands/sample.py:27-34): keep iff float(murmer(y, s) & (2^40 - 1)) / float(2^40 - 1) < p */
u64 zo_sample_d(double p, u64 seed, const u64* ys, const u64* yc, u64 ny, u64* ok, u64* oc) {
    const u64 M = 0xFFFFFFFFFFULL;
    u64 m = 0;
    for (u64 j = 0; j < ny; j++) {
        double u = (double)(zo_murmer(ys[j], seed) & M) / (double)M;
        if (u < p) { ok[m] = ys[j]; oc[m] = yc[j]; m++; }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * zot trim: zotmer/commands/trim.py
 * ---------------------------------------------------------------------------------------- */

/* trim.trim (zotmer/commands/trim.py:54-62): keep (x,f) iff f >= lo and (hi == 0 or f <= hi);
 * hi == 0 stands for the reference's C = None. */
u64 zo_trim(const u64* xs, const u64* cs, u64 n, u64 lo, u64 hi, u64* ox, u64* oc) {
    u64 m = 0;
    for (u64 i = 0; i < n; i++)
        if (cs[i] >= lo && (hi == 0 || cs[i] <= hi)) { ox[m] = xs[i]; oc[m] = cs[i]; m++; }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * on-disk vector codec: zotmer/library/codec64.py, zotmer/library/files.py
 * ---------------------------------------------------------------------------------------- */

static int bitlen(u64 x) { int n = 0; while (x) { x >>= 1; n++; } return n; }

/* codec64.encode / encoder (zotmer/library/codec64.py:42-120).  Effective rule of the
 * _lookup table (:33-40): a word takes the longest run of n <= 6 values whose widest member
 * fits in 60/n bits; word = n | sum(v_m << (4 + (60/n)*m)).  A value >= 2^60 has no code
 * (_lookup[0], the reference raises IndexError): returns -1.  words must hold n entries.
 * Returns the number of words. */
int64_t zo_codec64_encode(const u64* xs, u64 n, u64* words) {
    u64 nw = 0, i = 0;
    while (i < n) {
        int cnt = 0, mw = 0;
        while (cnt < 6 && i + cnt < n) {
            int w = bitlen(xs[i + cnt]);
            int mwx = w > mw ? w : mw;
            if (mwx > 60 / (cnt + 1)) break;
            mw = mwx; cnt++;
        }
        if (cnt == 0) return -1;
        int b = 60 / cnt;
        u64 v = 0;
        for (int m = cnt - 1; m >= 0; m--) v = (v << b) | xs[i + m];
        words[nw++] = (v << 4) | (u64)cnt;
        i += cnt;
    }
    return (int64_t)nw;
}

/* codec64.decode / decodeList (zotmer/library/codec64.py:122-151): tag n = w & 15, field
 * width _codes[n] = the value i for which 60 // i == n ... which for the tags the encoder
 * emits (1..6) is 60 / n.  Tags the encoder never writes (0, 7..15) follow the reference's
 * _codes dict (:28-31): _codes[b] = largest i with 60 // i == b; a tag with no entry
 * (0, 9, 11, 13, 14) is a KeyError there and -1 here.
 * out == NULL only counts.  Returns the number of values. */
static int codes_width(int tag) {
    int w = -1;
    for (int i = 1; i <= 60; i++) if (60 / i == tag) w = i;   /* last assignment wins */
    return w;
}
int64_t zo_codec64_decode(const u64* words, u64 nw, u64* out) {
    u64 n = 0;
    for (u64 j = 0; j < nw; j++) {
        u64 w = words[j];
        int m0 = (int)(w & 15);
        w >>= 4;
        int b = codes_width(m0);
        if (b < 0) return -1;               /* KeyError in the reference */
        u64 msk = (b >= 64) ? ~0ULL : ((1ULL << b) - 1);
        for (int m = 0; m < m0; m++) { if (out) out[n] = w & msk; n++; w >>= b; }
    }
    return (int64_t)n;
}

/* files.delta / deltaList (zotmer/library/files.py:85-98): d[i] = x[i] - x[i-1], x[-1] = 0. */
void zo_delta(const u64* xs, u64 n, u64* ds) {
    u64 p = 0;
    for (u64 i = 0; i < n; i++) { u64 x = xs[i]; ds[i] = x - p; p = x; }
}

/* files.undelta / undeltaList (zotmer/library/files.py:100-110). */
void zo_undelta(const u64* ds, u64 n, u64* xs) {
    u64 x = 0;
    for (u64 i = 0; i < n; i++) { x += ds[i]; xs[i] = x; }
}
