// setops.hip -- sorted-set algebra on (k-mer, count) arrays with merge-path partitioning:
//   K5  2-way union with summed counts   (merge.merge, zotmer/commands/merge.py:26-86; and the
//                                         merge half of kmerize.merge, commands/kmerize.py:41-132)
//   K6  k-way union-sum                  (mergeNinto, commands/merge.py:127-163, kmerize.py:269-304)
//                                         as a tree of K5 passes
//   K9  intersect count (a, b, c)        (dist.split, zotmer/library/dist.py:241-265)
//
// A partition kernel cuts the merged sequence of A and B into tiles of TILE outputs by binary
// search on the merge-path diagonals (ties: A before B).  A workgroup stages its A slice and B
// slice in LDS (one element of halo on each side), every thread finds its own diagonal in LDS
// and merges ITEMS elements serially.  With A first on ties an equal pair is always adjacent:
// the A element absorbs the B count and the B element is dropped, so the union-sum is fused
// into the merge; survivors are compacted through LDS and written as one contiguous run whose
// position comes from a decoupled look-back over the tiles.
//
// Algorithmic bytes: (8 + cb) per input element read, (8 + cb) per output element written
// (cb = 4 or 8 count bytes); K9 reads 8 per element and writes 24 bytes in total.
#include "internal.hpp"

namespace zk {

constexpr int MRG_BLOCK = 512;
constexpr int MRG_ITEMS = 8;
constexpr int MRG_TILE = MRG_BLOCK * MRG_ITEMS;
constexpr int MRG_NW = MRG_BLOCK / 64;

// part[t] = number of A elements among the first min(t*TILE, nA+nB) merged elements
// packb > 0: B holds (key << packb) | count words; only the key takes part in the comparisons (packa: the same for A)
__global__ void merge_partition_kernel(const u64* __restrict__ A, u64 nA, const u64* __restrict__ B, u64 nB, u64* __restrict__ part,
                                       u32 tiles, int packb, int packa) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tiles) return;
    u64 D = (u64)t * MRG_TILE;
    if (D > nA + nB) D = nA + nB;
    u64 lo = D > nB ? D - nB : 0, hi = D < nA ? D : nA;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((A[mid] >> packa) <= (B[D - mid - 1] >> packb)) lo = mid + 1; else hi = mid;
    }
    part[t] = lo;
}

template <typename CT>
struct MergeSmem {
    union {
        struct {
            // The A slice and the B slice of a tile add up to MRG_TILE elements, so they share one buffer:
            //   ka = keys      : [0] = left halo A[a0-1], [1 .. nAt] = the A slice, [nAt+1] = a spare slot that is read
            //                    (never used) when the A cursor stands at its end
            //   kb = keys+nAt+2: [0 .. nBt-1] = the B slice, [nBt] = right halo B[b1]
            // Half the LDS of two full-size buffers: six workgroups per CU instead of three.
            u64 keys[MRG_TILE + 4];
            CT cnts[MRG_TILE + 4];
        } in;
        struct {
            u64 k[MRG_TILE];
            CT c[MRG_TILE];
        } out;
    };
    u32 wtot[MRG_NW];
    u64 tile_excl;
    u32 ticket;
};

struct MergeState {
    u64* status; u32* ticket; u32 ticket_base; u32 epoch; u32* err; u64* d_total; u32 tiles;
    int packb;      // > 0: the B side is one array of (key << packb) | count words (cB is not read)
    int packa;      // > 0: so is the A side (cA is not read)
    int disjoint;   // 1: no key occurs in both lists (the caller's knowledge; checked)
};

// MODE 0: union with summed counts.  MODE 1: projection (project.project2, zotmer/commands/project.py:29-40):
// only the B entries whose key also occurs in A survive, with B's own count; A contributes no entry.
template <typename CT, int MODE>
__global__ __launch_bounds__(MRG_BLOCK) void union_sum_kernel(const u64* __restrict__ A, const CT* __restrict__ cA, u64 nA,
                                                              const u64* __restrict__ B, const CT* __restrict__ cB, u64 nB,
                                                              const u64* __restrict__ part, u64* __restrict__ ok,
                                                              CT* __restrict__ oc, u64 cap, u64* acgt_w, MergeState st) {
    __shared__ MergeSmem<CT> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const u64 a0 = part[tile], a1 = part[tile + 1];
    u64 d0 = (u64)tile * MRG_TILE, d1 = d0 + MRG_TILE;
    if (d1 > nA + nB) d1 = nA + nB;
    const u64 b0 = d0 - a0, b1 = d1 - a1;
    const int nAt = (int)(a1 - a0), nBt = (int)(b1 - b0);

    u64* const ka = sm.in.keys;
    u64* const kb = sm.in.keys + nAt + 2;
    CT* const ca = sm.in.cnts;
    CT* const cb = sm.in.cnts + nAt + 2;
    // stage: ka[1 + i] = A[a0 + i], ka[0] = A[a0 - 1]; kb[j] = B[b0 + j], kb[nBt] = B[b1].  One flat index space over both
    // slices, a fixed number of rounds, every load of a thread issued before the first LDS write: written as two loops over
    // the slices the loads came out one global round trip per iteration (load, wait, ds_write, next), up to sixteen in a row
    // -- that, not the merge, was the tile's time.
    {
        constexpr int R = (MRG_TILE + 2 + MRG_BLOCK - 1) / MRG_BLOCK;
        const u64* safe_k = nA ? A : B;
        const CT* safe_c = (nA || st.packb) ? cA : cB;          // (packa: cA is the caller's dummy, a readable word)
        u64 kv[R];
        CT cv[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int sl = tid + r * MRG_BLOCK;
            const bool isA = sl <= nAt;
            const u64 g = isA ? a0 + (u64)sl : b0 + (u64)(sl - nAt - 1);          // A side: element index + 1
            const bool ok = (sl < nAt + nBt + 2) && (isA ? g >= 1 : g < nB);
            const u64* pk = ok ? (isA ? A + (g - 1) : B + g) : safe_k;
            const CT* pc = (ok && (isA ? !st.packa : !st.packb)) ? (isA ? cA + (g - 1) : cB + g) : safe_c;
            kv[r] = *pk;
            cv[r] = *pc;
            if (st.packb && ok && !isA) {
                cv[r] = (CT)(kv[r] & ((1ull << st.packb) - 1ull));
                kv[r] >>= st.packb;
            }
            if (st.packa && ok && isA) {
                cv[r] = (CT)(kv[r] & ((1ull << st.packa) - 1ull));
                kv[r] >>= st.packa;
            }
            if (!ok) { kv[r] = 0; cv[r] = 0; }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int sl = tid + r * MRG_BLOCK;
            if (sl < nAt + nBt + 2) {
                const int at = (sl <= nAt) ? sl : sl + 1;       // the B side starts at keys + nAt + 2
                sm.in.keys[at] = kv[r];
                sm.in.cnts[at] = cv[r];
            }
        }
    }
    __syncthreads();
    const bool have_left = a0 > 0;
    const bool have_right = b1 < nB;

    // this thread's diagonal inside the tile
    const int total = nAt + nBt;
    int d = tid * MRG_ITEMS;
    if (d > total) d = total;
    int lo = d > nBt ? d - nBt : 0, hi = d < nAt ? d : nAt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ka[1 + mid] <= kb[d - mid - 1]) lo = mid + 1; else hi = mid;
    }
    int i = lo, j = d - lo;
    u64 rk[MRG_ITEMS];
    CT rc[MRG_ITEMS];
    u32 keep = 0;
#pragma unroll
    for (int s = 0; s < MRG_ITEMS; s++) {
        rk[s] = 0; rc[s] = 0;
        if (d + s < total) {
            const bool hasA = i < nAt, hasB = j < nBt;
            const u64 ak = ka[1 + i], bk = kb[j];
            if (hasA && (!hasB || ak <= bk)) {
                // the equal partner, if any, is the B cursor (possibly the right halo)
                const bool bvalid = (j < nBt) || have_right;
                CT c = ca[1 + i];
                if (MODE == 0) {
                    if (bvalid && bk == ak) {
                        const CT c2 = c + cb[j];
                        if (c2 < c) atomicOr(st.err, ZK_DERR_COUNT_OVERFLOW);
                        c = c2;
                    }
                    rk[s] = ak; rc[s] = c; keep |= 1u << s;
                }
                i++;
            } else {
                // dropped when the A element just before it (possibly the left halo) is equal
                const bool avalid = (i > 0) || have_left;
                const bool dup = avalid && ka[i] == bk;
                rk[s] = bk; rc[s] = cb[j];
                if (MODE == 0 ? !dup : dup) keep |= 1u << s;
                j++;
            }
        }
    }
    // compact: blocked arrangement -> exclusive scan of per-thread totals
    const u32 mine = (u32)__popc(keep);
    const u32 inc = wave_incl_scan_u32(mine);
    if (lane == 63) sm.wtot[wave] = inc;
    __syncthreads();           // also: everyone is done reading sm.in
    u32 wex = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MRG_NW; w++) { if (w < wave) wex += sm.wtot[w]; tot += sm.wtot[w]; }
    if (wave == 0) {
        // disjoint: the caller knows that no key is in both lists (a canonical list and its mirror image at odd K): every tile keeps all
        // its entries, its place in the output is its place in the merge -- no tile waits for another (the wait for the tiles before is
        // 3.5 of the kernel's 10.6 ms on config 2's two lists).  A tile that does drop an entry reports it (ZK_DERR_SHARED_KEY).
        u64 ex;
        if (st.disjoint) {
            ex = d0;
            if (lane == 0 && (u64)tot != d1 - d0) atomicOr(st.err, ZK_DERR_SHARED_KEY);
        } else ex = lookback_exclusive(st.status, tile, tot, st.epoch, st.err);
        if (lane == 0) {
            sm.tile_excl = ex;
            if (tile == st.tiles - 1) *st.d_total = ex + tot;
        }
    }
    u32 q = wex + inc - mine;
    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (int s = 0; s < MRG_ITEMS; s++) {
        if ((keep >> s) & 1u) {
            sm.out.k[q] = rk[s]; sm.out.c[q] = rc[s]; q++;
            const u32 b = (u32)(rk[s] & 3);
            const u64 c = (u64)rc[s];
            w0 += (b == 0) ? c : 0; w1 += (b == 1) ? c : 0; w2 += (b == 2) ? c : 0; w3 += (b == 3) ? c : 0;
        }
    }
    __syncthreads();
    const u64 base = sm.tile_excl;
    for (u32 s = tid; s < tot; s += MRG_BLOCK) {
        if (base + s < cap) { ok[base + s] = sm.out.k[s]; oc[base + s] = sm.out.c[s]; }
    }
    if (tid == 0 && tile == st.tiles - 1 && base + tot > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
    if (acgt_w) {
        // one partial row per tile (no contended atomics); column_sum_kernel adds them up
        __shared__ u64 scratch[MRG_NW];
        w0 = block_sum_u64(w0, scratch); w1 = block_sum_u64(w1, scratch);
        w2 = block_sum_u64(w2, scratch); w3 = block_sum_u64(w3, scratch);
        if (tid == 0) { u64* row = acgt_w + 4ull * tile; row[0] = w0; row[1] = w1; row[2] = w2; row[3] = w3; }
    }
}

// out[c] += sum over rows of in[row][c]; `cols` <= 8; the workgroups take slices of the rows (out is zeroed by the caller: one workgroup
// alone walked the 195 K rows of config 4's share in 0.38 ms)
__global__ void column_sum_kernel(const u64* __restrict__ in, u64 rows, int cols, u64* __restrict__ out) {
    __shared__ u64 scratch[16];
    for (int c = 0; c < cols; c++) {
        u64 s = 0;
        for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (u64)gridDim.x * blockDim.x) s += in[r * cols + c];
        s = block_sum_u64(s, scratch);
        if (threadIdx.x == 0 && s) atomicAdd((unsigned long long*)&out[c], (unsigned long long)s);
    }
}

// K9: number of common elements of two sorted unique arrays.  Workgroups stride over the tiles and
// add ONE number each at the end: a per-tile atomic on a single word was 4.4 of this kernel's 4.7 ms
// at 2 x 100 M keys (one address takes about 88 atomics per microsecond).
__global__ __launch_bounds__(MRG_BLOCK) void intersect_kernel(const u64* __restrict__ A, u64 nA, const u64* __restrict__ B, u64 nB,
                                                              const u64* __restrict__ part, u32 tiles, u64* __restrict__ n_common) {
    __shared__ u64 ka[MRG_TILE + 2];
    __shared__ u64 kb[MRG_TILE + 2];
    __shared__ u64 scratch[MRG_NW];
    const int tid = threadIdx.x;
    u64 total_hits = 0;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const u64 a0 = part[tile], a1 = part[tile + 1];
    u64 d0 = (u64)tile * MRG_TILE, d1 = d0 + MRG_TILE;
    if (d1 > nA + nB) d1 = nA + nB;
    const u64 b0 = d0 - a0, b1 = d1 - a1;
    const int nAt = (int)(a1 - a0), nBt = (int)(b1 - b0);
    {
        // all loads of a thread before its first LDS write (see union_sum_kernel)
        constexpr int R = (MRG_TILE + 1 + MRG_BLOCK - 1) / MRG_BLOCK;
        u64 kv[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int sl = tid + r * MRG_BLOCK;
            const bool isA = sl <= nAt;
            const u64 g = isA ? a0 + (u64)sl : b0 + (u64)(sl - nAt - 1);
            const bool ok = (sl < nAt + 1 + nBt) && (isA ? g >= 1 : true);
            kv[r] = *(ok ? (isA ? A + (g - 1) : B + g) : A);
            if (!ok) kv[r] = 0;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int sl = tid + r * MRG_BLOCK;
            if (sl <= nAt) ka[sl] = kv[r];
            else if (sl < nAt + 1 + nBt) kb[sl - nAt - 1] = kv[r];
        }
    }
    __syncthreads();
    const bool have_left = a0 > 0;
    const int total = nAt + nBt;
    int d = tid * MRG_ITEMS;
    if (d > total) d = total;
    int lo = d > nBt ? d - nBt : 0, hi = d < nAt ? d : nAt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ka[1 + mid] <= kb[d - mid - 1]) lo = mid + 1; else hi = mid;
    }
    int i = lo, j = d - lo;
    u32 hits = 0;
#pragma unroll
    for (int s = 0; s < MRG_ITEMS; s++) {
        if (d + s < total) {
            const bool hasA = i < nAt, hasB = j < nBt;
            if (hasA && (!hasB || ka[1 + i] <= kb[j])) i++;
            else {
                const bool avalid = (i > 0) || have_left;
                hits += (avalid && ka[i] == kb[j]) ? 1u : 0u;
                j++;
            }
        }
    }
    total_hits += hits;
    __syncthreads();        // the next tile restages ka / kb
    }
    total_hits = block_sum_u64(total_hits, scratch);
    if (tid == 0 && total_hits) atomicAdd(n_common, total_hits);
}

int column_sum(zk_ctx* c, const u64* rows, uint64_t n_rows, int cols, u64* out) {
    ZK_HIP(c, hipMemsetAsync(out, 0, sizeof(u64) * cols, c->stream));
    const u32 grid = (u32)(n_rows / 4096 + 1 < 256 ? n_rows / 4096 + 1 : 256);
    hipLaunchKernelGGL(column_sum_kernel, dim3(grid), dim3(1024), 0, c->stream, rows, (u64)n_rows, cols, out);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

static int make_partition(zk_ctx* c, const u64* A, u64 nA, const u64* B, u64 nB, u64** part, u32* tiles, int packb = 0, int packa = 0) {
    *tiles = (u32)div_up(nA + nB, MRG_TILE);
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)*tiles + 1), (void**)part));
    hipLaunchKernelGGL(merge_partition_kernel, dim3((u32)div_up((uint64_t)*tiles + 1, 256)), dim3(256), 0, c->stream, A, nA, B, nB,
                       *part, *tiles, packb, packa);
    ZK_HIP(c, hipGetLastError());
    return ZK_OK;
}

template <typename CT, int MODE>
static int union_sum_t(zk_ctx* c, const u64* A, const CT* cA, u64 nA, const u64* B, const CT* cB, u64 nB, u64* ok, CT* oc,
                       uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4], int packb = 0, int packa = 0, bool disjoint = false) {
    *n_out = 0;
    if (acgt_w) acgt_w[0] = acgt_w[1] = acgt_w[2] = acgt_w[3] = 0;
    if (nA + nB == 0) return ZK_OK;
    u64* part; u32 tiles;
    if (c->arena_off == 0) {          // a direct call: size the workspace for the partition AND the acgt rows before handing any out
        const uint64_t need = 40ull * (div_up(nA + nB, MRG_TILE) + 2) + 4096;
        ZK_TRY(arena_require(c, need, need));
    }
    ZK_TRY(make_partition(c, A, nA, B, nB, &part, &tiles, packb, packa));
    MergeState st;
    st.tiles = tiles;
    st.packb = packb;
    st.packa = packa;
    st.disjoint = disjoint ? 1 : 0;
    ZK_TRY(lookback_begin(c, tiles, tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9;
    u64* d_rows = nullptr;
    if (acgt_w) ZK_TRY(arena_alloc(c, 32ull * tiles, (void**)&d_rows));
    prof_begin(c, ZK_PROF_UNION, (packa ? 8 : 8 + sizeof(CT)) * nA + (packb ? 8 : 8 + sizeof(CT)) * nB);
    hipLaunchKernelGGL((union_sum_kernel<CT, MODE>), dim3(tiles), dim3(MRG_BLOCK), 0, c->stream, A, cA, nA, B, cB, nB, part, ok, oc,
                       (u64)cap, d_rows, st);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    if (acgt_w) {
        ZK_TRY(column_sum(c, d_rows, tiles, 4, c->d_scalars + 0));
    }
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(u64) * 16, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = c->h_scalars[9];
    prof_add_bytes(c, ZK_PROF_UNION, (8 + sizeof(CT)) * *n_out);          // the entries written (the launch was opened with the bytes read)
    if (acgt_w) for (int b = 0; b < 4; b++) acgt_w[b] = c->h_scalars[b];
    return check_device_error(c);
}

int union_sum(zk_ctx* c, const u64* A, const void* cA, u64 nA, const u64* B, const void* cB, u64 nB, u64* ok, void* oc,
              int count_bits, uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4], bool disjoint) {
    if (count_bits == 32)
        return union_sum_t<u32, 0>(c, A, (const u32*)cA, nA, B, (const u32*)cB, nB, ok, (u32*)oc, cap, n_out, acgt_w, 0, 0, disjoint);
    return union_sum_t<u64, 0>(c, A, (const u64*)cA, nA, B, (const u64*)cB, nB, ok, (u64*)oc, cap, n_out, acgt_w, 0, 0, disjoint);
}

int union_sum_packed_b(zk_ctx* c, const u64* A, const u32* cA, u64 nA, const u64* Bp, u64 nB, int pack, u64* ok, u32* oc, uint64_t cap,
                       uint64_t* n_out, bool disjoint) {
    return union_sum_t<u32, 0>(c, A, cA, nA, Bp, cA, nB, ok, oc, cap, n_out, nullptr, pack, 0, disjoint);
}

// both sides as (key << pack) | count words (the counted canonical list as the block dedupe leaves it, and its mirror image)
int union_sum_packed_ab(zk_ctx* c, const u64* Ap, u64 nA, const u64* Bp, u64 nB, int pack, u64* ok, u32* oc, uint64_t cap, uint64_t* n_out,
                        bool disjoint) {
    const u32* dummy = reinterpret_cast<const u32*>(nA ? Ap : Bp);          // a readable address for the count loads that are not made
    return union_sum_t<u32, 0>(c, Ap, dummy, nA, Bp, dummy, nB, ok, oc, cap, n_out, nullptr, pack, pack, disjoint);
}

// the reference's counts are not read in MODE 1, so the keys stand in for A's (absent) count array
int project(zk_ctx* c, const u64* ref, u64 n_ref, const u64* B, const u64* cB, u64 nB, u64* ok, u64* oc, uint64_t cap, uint64_t* n_out) {
    return union_sum_t<u64, 1>(c, ref, ref, n_ref, B, cB, nB, ok, oc, cap, n_out, nullptr);
}

int intersect_count(zk_ctx* c, const u64* A, u64 nA, const u64* B, u64 nB, uint64_t abc[3]) {
    abc[0] = 0; abc[1] = nA; abc[2] = nB;
    if (nA == 0 || nB == 0) return ZK_OK;
    u64* part; u32 tiles;
    ZK_TRY(make_partition(c, A, nA, B, nB, &part, &tiles));
    u64* d_n = c->d_scalars + 11;
    ZK_HIP(c, hipMemsetAsync(d_n, 0, sizeof(u64), c->stream));
    prof_begin(c, ZK_PROF_INTERSECT, 8 * (nA + nB));
    const u32 grid = tiles < (u32)c->num_cus * 8 ? tiles : (u32)c->num_cus * 8;
    hipLaunchKernelGGL(intersect_kernel, dim3(grid), dim3(MRG_BLOCK), 0, c->stream, A, nA, B, nB, part, tiles, d_n);
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 11, d_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    const uint64_t a = c->h_scalars[11];
    abc[0] = a; abc[1] = nA - a; abc[2] = nB - a;
    return ZK_OK;
}

}  // namespace zk
