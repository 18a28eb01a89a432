// comm.hip -- zk_comm_*: the multi-GPU seam of the C-ABI over RCCL.
//
// One process per GPU; a zk_ctx gets one RCCL communicator.  The data path has exactly two operations:
//   zk_all_to_all_v   -- the one exchange step of kmerize / merge / dist: piece r of this rank's partitioned
//                        table goes to rank r.  xGMI is a full mesh of point-to-point links, so this is one
//                        grouped ncclSend / ncclRecv per peer (one hop, per-link bound), straight from and into
//                        the caller's device arrays -- no staging copy -- in rounds of <= 256 MiB per message.
//   zk_allreduce_u64  -- scalars and small histograms (dist's (a, b, c), checksums, the splitter histogram).
// The reference has no communication of any kind (it is single-process Python); these entries exist so that a
// host that is not PyTorch can drive the 8-GPU path.  The id from zk_comm_unique_id (rank 0) reaches the other
// ranks by whatever the host has -- a file, MPI, or torch.distributed's store (zotmer_amd/parallel.py).
//
// RCCL is bound at run time (dlopen): libzotk.so itself does not depend on it, a single-GPU user never loads it,
// and inside a PyTorch process the copy PyTorch already mapped is the one that is used (same SONAME).
#include <dlfcn.h>
#include <string.h>

#include <vector>

#include <rccl/rccl.h>

#include "internal.hpp"

namespace zk {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

static Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.error.empty()) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // a copy that is already mapped (PyTorch's) wins
    for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.lib) break; }
    for (const char* n : names) { if (r.lib) break; r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
    if (!r.lib) { r.error = std::string("cannot load RCCL: ") + dlerror(); return &r; }
#define ZK_SYM(f)                                                         \
    r.f = (decltype(r.f))dlsym(r.lib, "nccl" #f);                         \
    if (!r.f) { r.error = "RCCL lacks nccl" #f; r.lib = nullptr; return &r; }
    ZK_SYM(GetUniqueId) ZK_SYM(CommInitRank) ZK_SYM(CommDestroy) ZK_SYM(GroupStart) ZK_SYM(GroupEnd) ZK_SYM(Send) ZK_SYM(Recv)
    ZK_SYM(AllReduce) ZK_SYM(GetErrorString)
#undef ZK_SYM
    return &r;
}

#define ZK_NCCL(c, call)                                                                                        \
    do {                                                                                                        \
        ncclResult_t r__ = (call);                                                                              \
        if (r__ != ncclSuccess)                                                                                 \
            return zk::fail((c), ZK_EHIP, "%s failed: %s (%s:%d)", #call, rccl()->GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

// The schedule of zk_all_to_all_v, as data: the messages rank `me` of `world` puts into round j, in the order it issues them.
// A pure function of the counts (no GPU, no RCCL), so that the part of the exchange a box with one GPU cannot run -- who talks to
// whom in which round, at which byte offsets -- is checked on the CPU for every world size (tests/test_comm_plan.py): over all
// ranks every send has exactly one receive of the same length in the same round, and the pieces are covered once.
static void a2a_plan(int world, int me, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off, const uint64_t* recv_cnt,
                     uint64_t eb, uint64_t chunk, bool self_loop, std::vector<zk_comm_op>* ops, uint64_t* rounds_out) {
    uint64_t biggest = 0;
    for (int p = 0; p < world; p++) {
        if (p == me && !self_loop) continue;
        if (send_cnt[p] * eb > biggest) biggest = send_cnt[p] * eb;
        if (recv_cnt[p] * eb > biggest) biggest = recv_cnt[p] * eb;
    }
    const uint64_t rounds = (biggest + chunk - 1) / chunk;
    for (uint64_t j = 0; j < rounds; j++) {
        for (int d = self_loop ? 0 : 1; d < world; d++) {
            // talk to (me + d) and (me - d) in the same step, so that every link is busy in both directions
            const int to = (me + d) % world, from = (me - d + world) % world;
            const uint64_t sb = send_cnt[to] * eb, rb = recv_cnt[from] * eb;
            if (j * chunk < sb) ops->push_back(zk_comm_op{0, to, j, send_off[to] * eb + j * chunk, sb - j * chunk < chunk ? sb - j * chunk : chunk});
            if (j * chunk < rb) ops->push_back(zk_comm_op{1, from, j, recv_off[from] * eb + j * chunk, rb - j * chunk < chunk ? rb - j * chunk : chunk});
        }
    }
    if (rounds_out) *rounds_out = rounds;
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_comm_plan(int world, int rank, const uint64_t* send_off, const uint64_t* send_cnt, const uint64_t* recv_off, const uint64_t* recv_cnt,
                 int elem_bytes, uint64_t chunk_bytes, int self_loop, zk_comm_op* ops, uint64_t cap, uint64_t* n_ops) {
    if (world < 1 || rank < 0 || rank >= world || !send_off || !send_cnt || !recv_off || !recv_cnt || elem_bytes < 1 || !n_ops) return ZK_EINVAL;
    std::vector<zk_comm_op> v;
    a2a_plan(world, rank, send_off, send_cnt, recv_off, recv_cnt, (uint64_t)elem_bytes, chunk_bytes ? chunk_bytes : (256ull << 20), self_loop != 0, &v, nullptr);
    *n_ops = v.size();
    if (ops) for (uint64_t i = 0; i < v.size() && i < cap; i++) ops[i] = v[i];
    return ZK_OK;
}

int zk_comm_unique_id(uint8_t id[ZK_COMM_ID_BYTES]) {
    Rccl* r = rccl();
    if (!r->lib || !id) return ZK_EHIP;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return ZK_EHIP;
    static_assert(sizeof(u) == ZK_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return ZK_OK;
}

int zk_comm_init(zk_ctx* c, int world, int rank, const uint8_t id[ZK_COMM_ID_BYTES]) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(c, ZK_EINVAL, "zk_comm_init: bad world / rank / id");
    if (c->comm) return fail(c, ZK_EINVAL, "zk_comm_init: the context already has a communicator");
    Rccl* r = rccl();
    if (!r->lib) return fail(c, ZK_EHIP, "%s", r->error.c_str());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    ZK_NCCL(c, r->CommInitRank(&comm, world, u, rank));
    c->comm = comm; c->comm_world = world; c->comm_rank = rank;
    return ZK_OK;
}

int zk_comm_destroy(zk_ctx* c) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy((ncclComm_t)c->comm);
        c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    }
    return ZK_OK;
}

int zk_comm_info(zk_ctx* c, int* world, int* rank) {
    if (!c) return ZK_EINVAL;
    if (world) *world = c->comm ? c->comm_world : 1;
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    return ZK_OK;
}

int zk_all_to_all_v(zk_ctx* c, const void* d_send, const uint64_t* send_off, const uint64_t* send_cnt, void* d_recv,
                    const uint64_t* recv_off, const uint64_t* recv_cnt, int elem_bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (!c->comm) return fail(c, ZK_EINVAL, "zk_all_to_all_v: no communicator (call zk_comm_init)");
    if (!send_off || !send_cnt || !recv_off || !recv_cnt || elem_bytes < 1) return fail(c, ZK_EINVAL, "zk_all_to_all_v: bad argument");
    Rccl* r = rccl();
    const int W = c->comm_world, me = c->comm_rank;
    const uint64_t eb = (uint64_t)elem_bytes;
    if (send_cnt[me] != recv_cnt[me]) return fail(c, ZK_EINVAL, "zk_all_to_all_v: the piece kept locally has two sizes");
    // The piece a rank keeps is a device copy -- unless ZK_TUNE_COMM_SELF_LOOP sends it through RCCL as well (a grouped ncclSend /
    // ncclRecv to the rank itself, in the same rounds as every other piece): that is how the loop below runs on a box with one GPU.
    const bool self_loop = c->comm_self_loop != 0;
    if (send_cnt[me] && !self_loop)
        ZK_HIP(c, hipMemcpyAsync((char*)d_recv + recv_off[me] * eb, (const char*)d_send + send_off[me] * eb, send_cnt[me] * eb,
                                 hipMemcpyDeviceToDevice, c->stream));
    if (W == 1 && !self_loop) return ZK_OK;
    const uint64_t chunk = c->comm_chunk_bytes ? c->comm_chunk_bytes : (256ull << 20);     // bytes per message and round
    std::vector<zk_comm_op> ops;
    uint64_t rounds = 0;
    a2a_plan(W, me, send_off, send_cnt, recv_off, recv_cnt, eb, chunk, self_loop, &ops, &rounds);
    size_t i = 0;
    for (uint64_t j = 0; j < rounds; j++) {
        ZK_NCCL(c, r->GroupStart());
        for (; i < ops.size() && ops[i].round == j; i++) {
            const zk_comm_op& o = ops[i];
            if (o.recv) ZK_NCCL(c, r->Recv((char*)d_recv + o.offset, o.bytes, ncclUint8, o.peer, (ncclComm_t)c->comm, c->stream));
            else ZK_NCCL(c, r->Send((const char*)d_send + o.offset, o.bytes, ncclUint8, o.peer, (ncclComm_t)c->comm, c->stream));
        }
        ZK_NCCL(c, r->GroupEnd());
    }
    return ZK_OK;
}

int zk_allreduce_u64(zk_ctx* c, uint64_t* vals, uint64_t n, int op) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (!c->comm) return fail(c, ZK_EINVAL, "zk_allreduce_u64: no communicator (call zk_comm_init)");
    if (n == 0) return ZK_OK;
    if (!vals || (op != ZK_REDUCE_SUM && op != ZK_REDUCE_MAX)) return fail(c, ZK_EINVAL, "zk_allreduce_u64: bad argument");
    if (c->comm_world == 1 && !c->comm_self_loop) return ZK_OK;          // (ZK_TUNE_COMM_SELF_LOOP: through ncclAllReduce with one rank as well)
    arena_reset(c);
    u64* d;
    ZK_TRY(arena_alloc(c, 8 * n, (void**)&d));
    ZK_HIP(c, hipMemcpyAsync(d, vals, 8 * n, hipMemcpyHostToDevice, c->stream));
    ZK_NCCL(c, rccl()->AllReduce(d, d, n, ncclUint64, op == ZK_REDUCE_SUM ? ncclSum : ncclMax, (ncclComm_t)c->comm, c->stream));
    ZK_HIP(c, hipMemcpyAsync(vals, d, 8 * n, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

}  // extern "C"
