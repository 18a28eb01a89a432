// kway.hip -- K6: union-sum of up to 16 sorted (k-mer, count) lists in ONE pass over the data.
//
// Replaces mergeNinto's heap + dict (zotmer/commands/merge.py:127-163, commands/kmerize.py:269-304), which merges all k streams
// at once; the tree of 2-way passes in pipeline.hip::merge_many reads and writes every pair ceil(log2 k) times instead (three
// levels for the eight sets a GPU holds in BASELINE config 4: 0.29 of the HBM peak).  Here every pair is read once:
//   * a SAMPLE -- every S-th key of every list, and each list's last -- is sorted; every T-th key of the sorted sample is a
//     splitter.  Between two splitters a list holds fewer than (its sample points there + 1) * S elements, so a tile -- the
//     elements of all lists in (splitter j - 1, splitter j] -- holds at most (T + k) * S of them: it fits LDS by construction, and
//     equal keys of different lists never straddle a tile (the cut is by VALUE: per list an upper bound by binary search);
//   * a workgroup stages its tile's k runs in LDS and merges them pairwise, ceil(log2 k) rounds in place: per round a thread finds
//     its diagonal of the pair its outputs fall into by binary search and merges four elements serially (merge path, as setops.hip
//     does for two lists -- but the rounds cost LDS traffic, not HBM traffic);
//   * equal keys are now neighbours: the first of a run adds up the counts of the rest, the survivors are compacted and leave as one
//     contiguous piece whose place comes from a decoupled look-back over the tiles (common.hpp).
// Algorithmic bytes: (8 + cb) per input pair read, (8 + cb) per output pair written, cb = 4 or 8 count bytes.
#include "internal.hpp"

namespace zk {

constexpr int KW_MAX = 16;                    // lists per pass
constexpr int KW_BLOCK = 512, KW_ITEMS = 4, KW_CAP = KW_BLOCK * KW_ITEMS;          // elements a tile can hold
constexpr int KW_S = 64;                      // one sample point per S elements of a list
constexpr int KW_NW = KW_BLOCK / 64;

struct KwayLists {
    const u64* keys[KW_MAX];
    const void* cnts[KW_MAX];
    u64 n[KW_MAX];
    u64 sbase[KW_MAX + 1];          // where list i's sample points start in the sample array
    int k;
};

// sample[sbase[i] + j] = keys_i[min((j + 1) * S, n_i) - 1]: every S-th key and the list's last one
__global__ void kway_sample_kernel(KwayLists L, u64* __restrict__ samp) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.sbase[L.k]) return;
    int i = 0;
    while (i + 1 < L.k && t >= L.sbase[i + 1]) i++;
    const u64 j = t - L.sbase[i];
    u64 at = (j + 1) * KW_S;
    if (at > L.n[i]) at = L.n[i];
    samp[t] = L.keys[i][at - 1];
}

// bounds[j][i], j = 0 .. tiles: list i's elements of tile j are [bounds[j][i], bounds[j + 1][i]); the cut after tile j is the number of
// list i's keys <= sorted_sample[(j + 1) * T - 1] (an upper bound by binary search); the last cut is the list's end
__global__ void kway_bounds_kernel(KwayLists L, const u64* __restrict__ sorted_samp, u64 m, u32 T, u32 tiles, u64* __restrict__ bounds) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ((u64)tiles + 1) * L.k) return;
    const u32 j = (u32)(t / L.k);
    const int i = (int)(t % L.k);
    u64 b;
    if (j == 0) b = 0;
    else if (j == tiles || (u64)j * T - 1 >= m) b = L.n[i];
    else {
        const u64 s = sorted_samp[(u64)j * T - 1];
        u64 lo = 0, hi = L.n[i];
        const u64* kk = L.keys[i];
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (kk[mid] <= s) lo = mid + 1; else hi = mid;
        }
        b = lo;
    }
    bounds[(u64)j * L.k + i] = b;
}

struct KwayState {
    u64* status; u32* ticket; u32 ticket_base; u32 epoch; u32* err; u64* d_total; u32 tiles;
#ifdef ZK_PHASES
    int dbg_nolook;
#endif
};

// ONE buffer: a round's outputs wait in registers until every thread has read its inputs (two barriers a round instead of one) --
// with two buffers (64 KB) a CU held two workgroups of four waves and the serial merges had nothing to hide their LDS round trips
// behind: 13.2 ms for the eight sets of config 4 against 8.6 ms for the three levels of 2-way passes.
template <typename CT>
struct KwaySmem {
    u64 k[KW_CAP];
    CT c[KW_CAP];
    u32 roff[KW_MAX + 1];          // where the runs start in the buffer (run r of the current round: [roff[r << round], roff[(r + 1) << round]))
    u64 glo[KW_MAX];               // where the tile's piece of list i starts in the list
    u32 wtot[KW_NW];
    u64 acc[4];
    u64 tile_excl;
    u32 ticket;
};

// KT: lists rounded up (4, 8, 16): how far the search for an element's list is unrolled
template <typename CT, int KT>
__global__ __launch_bounds__(KW_BLOCK) void kway_merge_kernel(KwayLists L, const u64* __restrict__ bounds, u64* __restrict__ ok, CT* __restrict__ oc,
                                                              u64 cap, u64* acgt_rows, KwayState st) {
    __shared__ KwaySmem<CT> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 tile = take_ticket(st.ticket, &sm.ticket) - st.ticket_base;
    const int k = L.k;
    if (tid < KW_MAX) {
        // lane i: list i's piece of the tile; the runs' places by a scan over the 16 lanes
        u64 lo = 0, hi = 0;
        if (tid < k) { lo = bounds[(u64)tile * k + tid]; hi = bounds[((u64)tile + 1) * k + tid]; }
        const u32 len = (u32)(hi - lo);
        u32 inc = len;
#pragma unroll
        for (int o = 1; o < KW_MAX; o <<= 1) { const u32 t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
        sm.glo[tid] = lo;
        sm.roff[tid] = inc - len;
        if (tid == KW_MAX - 1) sm.roff[KW_MAX] = inc;
        if (tid < 4) sm.acc[tid] = 0;
    }
    __syncthreads();
    u32 total = sm.roff[KW_MAX];
    if (total > (u32)KW_CAP) {          // cannot happen (see the bound above); never write past the buffers
        if (tid == 0) atomicOr(st.err, ZK_DERR_CAPACITY);
        total = KW_CAP;
    }
    // ---- stage: every load of a thread issued before its first LDS write ---------------------------------------------------
    u64 kv[KW_ITEMS];
    CT cv[KW_ITEMS];
#pragma unroll
    for (int r = 0; r < KW_ITEMS; r++) {
        const u32 e = (u32)tid + r * KW_BLOCK;
        kv[r] = 0; cv[r] = 0;
        if (e < total) {
            int i = 0;
#pragma unroll
            for (int q = 1; q < KT; q++) i += e >= sm.roff[q] ? 1 : 0;
            const u32 at = e - sm.roff[i];
            const u64 g = sm.glo[i] + at;
            kv[r] = L.keys[i][g];
            cv[r] = reinterpret_cast<const CT*>(L.cnts[i])[g];
        }
    }
#pragma unroll
    for (int r = 0; r < KW_ITEMS; r++) {
        const u32 e = (u32)tid + r * KW_BLOCK;
        if (e < total) sm.k[e] = kv[r];
    }
    __syncthreads();
    // ---- merge rounds: runs (2p, 2p + 1) of width `w` lists each become run p of width 2w; a pair's output takes its inputs' span.
    // A round's outputs wait in registers until every thread has read its inputs (one buffer, two barriers a round: with two buffers,
    // 64 KB, a CU held two workgroups of four waves and nothing hid the serial merges' LDS round trips: 13.2 ms; so does ranking every
    // element by KT - 1 binary searches in the other runs instead of merging: 256 LDS reads a thread, 13.2 ms as well).
    {
        const u64* sk = sm.k;
#pragma unroll
        for (int r = 0; r < KW_ITEMS; r++) {
            const u32 e = (u32)tid + r * KW_BLOCK;
            if (e < total) sm.c[e] = cv[r];
        }
        __syncthreads();
        const CT* sc = sm.c;
        for (int w = 1; w < k; w <<= 1) {
            u64 mk[KW_ITEMS];
            CT mc[KW_ITEMS];
            u32 o = (u32)tid * KW_ITEMS;
            if (o < total) {
                // the pair this thread's first output falls into
                int p = 0;
                while (((p + 1) * 2 * w) < KW_MAX && o >= sm.roff[(p + 1) * 2 * w]) p++;
                u32 a0 = sm.roff[p * 2 * w], a1 = sm.roff[min(p * 2 * w + w, KW_MAX)], b1 = sm.roff[min((p + 1) * 2 * w, KW_MAX)];
                // its diagonal inside the pair
                const u32 d = o - a0, la = a1 - a0, lb = b1 - a1;
                u32 lo = d > lb ? d - lb : 0u, hi = d < la ? d : la;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    if (sk[a0 + mid] <= sk[a1 + (d - mid - 1)]) lo = mid + 1; else hi = mid;
                }
                u32 i = a0 + lo, j = a1 + (d - lo);
#pragma unroll
                for (int s2 = 0; s2 < KW_ITEMS; s2++, o++) {
                    mk[s2] = 0; mc[s2] = 0;
                    if (o >= total) continue;
                    if (o >= b1) {          // into the next pair, at its beginning
                        p++;
                        a0 = b1; a1 = sm.roff[min(p * 2 * w + w, KW_MAX)]; b1 = sm.roff[min((p + 1) * 2 * w, KW_MAX)];
                        i = a0; j = a1;
                        while (b1 == a0 && p * 2 * w < KW_MAX) {          // (empty pairs)
                            p++;
                            a0 = b1; a1 = sm.roff[min(p * 2 * w + w, KW_MAX)]; b1 = sm.roff[min((p + 1) * 2 * w, KW_MAX)];
                            i = a0; j = a1;
                        }
                    }
                    const bool hasA = i < a1, hasB = j < b1;
                    const u64 ak = sk[hasA ? i : a0], bk = sk[hasB ? j : a0];
                    const bool takeA = hasA && (!hasB || ak <= bk);
                    mk[s2] = takeA ? ak : bk;
                    mc[s2] = takeA ? sc[i] : sc[hasB ? j : a0];
                    if (takeA) i++; else j++;
                }
            }
            __syncthreads();          // every thread has read its inputs
            {
                const u32 o0 = (u32)tid * KW_ITEMS;
#pragma unroll
                for (int s2 = 0; s2 < KW_ITEMS; s2++) if (o0 + s2 < total) { sm.k[o0 + s2] = mk[s2]; sm.c[o0 + s2] = mc[s2]; }
            }
            __syncthreads();
        }
    }
    const u64* sk = sm.k;
    const CT* sc = sm.c;
    // ---- equal keys are neighbours: the first of a run takes the counts of the rest; survivors compacted ----------------------------
    u64 rk[KW_ITEMS];
    CT rc[KW_ITEMS];
    u32 keep = 0;
    {
        const u32 o0 = (u32)tid * KW_ITEMS;
#pragma unroll
        for (int s = 0; s < KW_ITEMS; s++) {
            const u32 o = o0 + s;
            rk[s] = 0; rc[s] = 0;
            if (o < total) {
                const u64 x = sk[o];
                if (o == 0 || sk[o - 1] != x) {
                    CT sum = sc[o];
                    for (u32 q = o + 1; q < total && sk[q] == x; q++) {
                        const CT s2 = sum + sc[q];
                        if (s2 < sum) atomicOr(st.err, ZK_DERR_COUNT_OVERFLOW);
                        sum = s2;
                    }
                    rk[s] = x; rc[s] = sum; keep |= 1u << s;
                }
            }
        }
    }
    const u32 mine = (u32)__popc(keep);
    const u32 inc = wave_incl_scan_u32(mine);
    if (lane == 63) sm.wtot[wave] = inc;
    __syncthreads();           // also: everyone is done reading the merged buffer
    u32 wex = 0, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < KW_NW; w2++) { if (w2 < wave) wex += sm.wtot[w2]; tot += sm.wtot[w2]; }
    if (wave == 0) {
#ifdef ZK_PHASES          // measurement (ZK_KWAY_NOLOOK=1; the result has gaps): what the wait for the tiles before costs
        const u64 ex = st.dbg_nolook ? (u64)tile * 1400ull : lookback_exclusive(st.status, tile, tot, st.epoch, st.err);
#else
        const u64 ex = lookback_exclusive(st.status, tile, tot, st.epoch, st.err);
#endif
        if (lane == 0) {
            sm.tile_excl = ex;
            if (tile == st.tiles - 1) *st.d_total = ex + tot;
        }
    }
    u64* dk = sm.k;          // (every thread is past the barrier above: the merged buffer has been read)
    CT* dc = sm.c;
    u32 q = wex + inc - mine;
    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (int s = 0; s < KW_ITEMS; s++) {
        if ((keep >> s) & 1u) {
            dk[q] = rk[s]; dc[q] = rc[s]; q++;
            const u32 b = (u32)(rk[s] & 3);
            const u64 cc = (u64)rc[s];
            w0 += (b == 0) ? cc : 0; w1 += (b == 1) ? cc : 0; w2 += (b == 2) ? cc : 0; w3 += (b == 3) ? cc : 0;
        }
    }
    if (acgt_rows) {
        w0 = wave_sum_u64(w0); w1 = wave_sum_u64(w1); w2 = wave_sum_u64(w2); w3 = wave_sum_u64(w3);
        if (lane == 0) {
            atomicAdd((unsigned long long*)&sm.acc[0], (unsigned long long)w0); atomicAdd((unsigned long long*)&sm.acc[1], (unsigned long long)w1);
            atomicAdd((unsigned long long*)&sm.acc[2], (unsigned long long)w2); atomicAdd((unsigned long long*)&sm.acc[3], (unsigned long long)w3);
        }
    }
    __syncthreads();
    const u64 base = sm.tile_excl;
    for (u32 s = tid; s < tot; s += KW_BLOCK) {
        if (base + s < cap) { ok[base + s] = dk[s]; oc[base + s] = dc[s]; }
    }
    if (tid == 0 && tile == st.tiles - 1 && base + tot > cap) atomicOr(st.err, ZK_DERR_CAPACITY);
    if (acgt_rows && tid < 4) acgt_rows[4ull * tile + tid] = sm.acc[tid];
}

// lists[0 .. k), 2 <= k <= KW_MAX, each ascending and without duplicates -> their union with summed counts
int kway_union_sum(zk_ctx* c, int k, const u64* const* keys, const void* const* cnts, const uint64_t* ns, u64* out_k, void* out_c, int count_bits,
                   uint64_t cap, uint64_t* n_out, uint64_t acgt_w[4]) {
    *n_out = 0;
    if (acgt_w) acgt_w[0] = acgt_w[1] = acgt_w[2] = acgt_w[3] = 0;
    if (k < 1 || k > KW_MAX) return fail(c, ZK_EINTERNAL, "kway_union_sum: %d lists", k);
    KwayLists L = {};
    L.k = k;
    uint64_t m = 0, total = 0;
    for (int i = 0; i < k; i++) {
        L.keys[i] = keys[i]; L.cnts[i] = cnts[i]; L.n[i] = ns[i];
        L.sbase[i] = m;
        m += div_up(ns[i], KW_S);
        total += ns[i];
    }
    L.sbase[k] = m;
    if (total == 0) return ZK_OK;
    // splitters: every T-th key of the sorted sample; a tile then holds at most (T + k) * S <= KW_CAP elements
    const u32 T = (u32)(KW_CAP / KW_S - k);
    const u32 tiles = (u32)(m / T + 1);
    u64 *samp, *salt, *bounds, *rows = nullptr;
    ZK_TRY(arena_alloc(c, sizeof(u64) * m, (void**)&samp));
    ZK_TRY(arena_alloc(c, sizeof(u64) * m, (void**)&salt));
    ZK_TRY(arena_alloc(c, sizeof(u64) * ((uint64_t)tiles + 1) * k, (void**)&bounds));
    if (acgt_w) ZK_TRY(arena_alloc(c, sizeof(u64) * 4 * (uint64_t)tiles, (void**)&rows));
    hipLaunchKernelGGL(kway_sample_kernel, dim3((u32)div_up(m, 256)), dim3(256), 0, c->stream, L, samp);
    ZK_HIP(c, hipGetLastError());
    u64* sorted = nullptr;
    ZK_TRY(sort_keys(c, samp, salt, m, 64, &sorted));
    hipLaunchKernelGGL(kway_bounds_kernel, dim3((u32)div_up(((uint64_t)tiles + 1) * k, 256)), dim3(256), 0, c->stream, L, (const u64*)sorted, (u64)m, T, tiles,
                       bounds);
    ZK_HIP(c, hipGetLastError());
    KwayState st = {};
    ZK_TRY(lookback_begin(c, tiles, tiles, &st.epoch, &st.ticket_base));
    st.status = c->status; st.ticket = c->d_ticket; st.err = c->d_err; st.d_total = c->d_scalars + 9; st.tiles = tiles;
#ifdef ZK_PHASES
    st.dbg_nolook = getenv("ZK_KWAY_NOLOOK") ? atoi(getenv("ZK_KWAY_NOLOOK")) : 0;
#endif
    const uint64_t cb = (uint64_t)count_bits / 8;
    prof_begin(c, ZK_PROF_UNION, (8 + cb) * total);
#define ZK_KW(CT, KT) hipLaunchKernelGGL((kway_merge_kernel<CT, KT>), dim3(tiles), dim3(KW_BLOCK), 0, c->stream, L, (const u64*)bounds, out_k, (CT*)out_c, (u64)cap, rows, st)
    if (count_bits == 32) { if (k <= 4) ZK_KW(u32, 4); else if (k <= 8) ZK_KW(u32, 8); else ZK_KW(u32, 16); }
    else { if (k <= 4) ZK_KW(u64, 4); else if (k <= 8) ZK_KW(u64, 8); else ZK_KW(u64, 16); }
#undef ZK_KW
    prof_end(c);
    ZK_HIP(c, hipGetLastError());
    if (acgt_w) ZK_TRY(column_sum(c, rows, tiles, 4, c->d_scalars + 0));
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(u64) * 16, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *n_out = c->h_scalars[9];
    prof_add_bytes(c, ZK_PROF_UNION, (8 + cb) * *n_out);
    if (acgt_w) for (int b = 0; b < 4; b++) acgt_w[b] = c->h_scalars[b];
    return check_device_error(c);
}

}  // namespace zk
