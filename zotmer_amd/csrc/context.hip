// context.hip -- zk_ctx: device, stream, workspace arena, look-back state, error reporting,
// and the buffer-management entry points of the C-ABI (include/zotk.h).
#include <stdarg.h>
#include <string.h>

#include "internal.hpp"
#include <vector>

namespace zk {

int fail(zk_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->last_error = buf;
    return code;
}

void arena_reset(zk_ctx* c) { c->arena_off = 0; }

int arena_alloc(zk_ctx* c, uint64_t bytes, void** p) {
    const uint64_t need = (bytes + 255) & ~255ull;
    if (c->arena_off + need > c->arena_size) {
        if (c->arena_off != 0)
            return fail(c, ZK_ENOMEM, "workspace too small: %llu bytes in use, %llu more needed, %llu reserved (call zk_reserve)",
                        (unsigned long long)c->arena_off, (unsigned long long)need, (unsigned long long)c->arena_size);
        // nothing handed out yet in this call: safe to regrow
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        if (c->arena) { ZK_HIP(c, hipFree(c->arena)); c->arena = nullptr; c->arena_size = 0; }
        const uint64_t floor_bytes = 64ull << 20;       // small follow-up allocations of the same call must fit too
        const uint64_t want = need + (need < floor_bytes ? floor_bytes : need / 16);
        ZK_HIP(c, hipMalloc((void**)&c->arena, want));
        c->arena_size = want;
    }
    *p = c->arena + c->arena_off;
    c->arena_off += need;
    return ZK_OK;
}

// Make the arena at least `want` bytes (clamped to what the device can give); call before the
// first arena_alloc of an API call.  `must` is the part the call cannot run without.
int arena_require(zk_ctx* c, uint64_t want, uint64_t must) {
    if (want <= c->arena_size) return ZK_OK;
    if (c->arena_off != 0) return fail(c, ZK_EINTERNAL, "arena_require after arena_alloc");
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    size_t f = 0, t = 0;
    ZK_HIP(c, hipMemGetInfo(&f, &t));
    uint64_t avail = (uint64_t)((double)(f + c->arena_size) * 0.94);
    uint64_t target = want < avail ? want : avail;
    if (target < must) return fail(c, ZK_ENOMEM, "needs %llu bytes of workspace, the device can give %llu",
                                   (unsigned long long)must, (unsigned long long)avail);
    if (target <= c->arena_size) return ZK_OK;
    if (c->arena) { ZK_HIP(c, hipFree(c->arena)); c->arena = nullptr; c->arena_size = 0; }
    target = (target + 255) & ~255ull;
    hipError_t e = hipMalloc((void**)&c->arena, target);
    if (e != hipSuccess) return fail(c, ZK_ENOMEM, "hipMalloc(%llu) for the workspace failed: %s", (unsigned long long)target, hipGetErrorString(e));
    c->arena_size = target;
    return ZK_OK;
}

int aux_require(zk_ctx* c, uint64_t bytes, char** p) {
    if (bytes > c->aux_size) {
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        if (c->aux) { ZK_HIP(c, hipFree(c->aux)); c->aux = nullptr; c->aux_size = 0; }
        const uint64_t want = (bytes + bytes / 16 + 4095) & ~4095ull;
        hipError_t e = hipMalloc((void**)&c->aux, want);
        if (e != hipSuccess) return fail(c, ZK_ENOMEM, "hipMalloc(%llu) for the mirror buffers failed: %s", (unsigned long long)want, hipGetErrorString(e));
        c->aux_size = want;
    }
    *p = c->aux;
    return ZK_OK;
}

void prof_begin(zk_ctx* c, int tag, uint64_t bytes) {
    if (!c->profile) return;
    zk_ctx::ProfRec r;
    r.tag = tag; r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { c->profile = false; return; }
    (void)hipEventRecord(r.a, c->stream);
    c->prof.push_back(r);
}
void prof_end(zk_ctx* c) {
    if (!c->profile || c->prof.empty()) return;
    (void)hipEventRecord(c->prof.back().b, c->stream);
}
// the bytes a launch wrote, when only its result says how many (distinct keys of a dedupe, entries of a union): added to the
// record of that tag opened last
void prof_add_bytes(zk_ctx* c, int tag, uint64_t bytes) {
    if (!c->profile) return;
    for (size_t i = c->prof.size(); i-- > 0;)
        if (c->prof[i].tag == tag) { c->prof[i].bytes += bytes; return; }
}

int lookback_begin(zk_ctx* c, uint64_t words, uint32_t tiles, u32* epoch, u32* ticket_base) {
    if (words > c->status_words) {
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        if (c->status) { ZK_HIP(c, hipFree(c->status)); c->status = nullptr; c->status_words = 0; }
        const uint64_t w = words + words / 8 + 1024;
        ZK_HIP(c, hipMalloc((void**)&c->status, w * sizeof(u64)));
        c->status_words = w;
        c->epoch = 0;
    }
    if (c->epoch == 0 || c->epoch >= 31) {
        ZK_HIP(c, hipMemsetAsync(c->status, 0, c->status_words * sizeof(u64), c->stream));
        c->epoch = 0;
    }
    c->epoch++;
    *epoch = c->epoch;
    *ticket_base = c->ticket_base;
    c->ticket_base += tiles;   // wraps with the device counter
    return ZK_OK;
}

int part16_begin(zk_ctx* c, uint64_t words, unsigned short** out) {
    if (words > c->part16_words) {
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        if (c->part16) { ZK_HIP(c, hipFree(c->part16)); c->part16 = nullptr; c->part16_words = 0; }
        const uint64_t w = words + words / 8 + 4096;
        ZK_HIP(c, hipMalloc((void**)&c->part16, w * sizeof(unsigned short)));
        c->part16_words = w;
    }
    ZK_HIP(c, hipMemsetAsync(c->part16, 0, words * sizeof(unsigned short), c->stream));
    *out = c->part16;
    return ZK_OK;
}

int check_device_error(zk_ctx* c) {
    u32 e = 0;
    ZK_HIP(c, hipMemcpyAsync(&e, c->d_err, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    if (e == 0) return ZK_OK;
    ZK_HIP(c, hipMemsetAsync(c->d_err, 0, sizeof(u32), c->stream));
    if (e & ZK_DERR_SPIN_TIMEOUT) {
        // the ticket counter and status words are in an unknown state: start over
        (void)hipMemsetAsync(c->d_ticket, 0, sizeof(u32), c->stream);
        c->ticket_base = 0;
        c->epoch = 0;
        // bits 8.. say which wait gave up: 0x01-0x20 radix_sort.hip in file order (array-pass walk, the two segmented-sort waits,
        // scanner waiting for counts, scanner wave hand-off, tile waiting for offsets), 0x100 codec.hip parse state,
        // 0x200 common.hpp look-back
        return fail(c, ZK_EINTERNAL, "device look-back spin limit reached (kernel bug or lost workgroup; wait sites 0x%x)", e >> 8);
    }
    if (e & ZK_DERR_RANGE) return fail(c, ZK_ERANGE, "a value (or k-mer delta) >= 2^60 cannot be stored in the codec64 format");
    if (e & ZK_DERR_BAD_TAG) return fail(c, ZK_ERANGE, "corrupt codec64 stream (unknown tag)");
    if (e & ZK_DERR_CAPACITY) return fail(c, ZK_ENOSPC, "output does not fit the capacity given");
    if (e & ZK_DERR_COUNT_OVERFLOW) return fail(c, ZK_EOVERFLOW, "a k-mer count does not fit the count type");
    if (e & ZK_DERR_SHARED_KEY) return fail(c, ZK_EINTERNAL, "two lists merged as disjoint share a key (the strands of a list that is not canonical?)");
    if (e & ZK_DERR_MISMATCH) return fail(c, ZK_EINTERNAL, "the first sort pass and the histogram before it disagree on the keys of a stream range");
    return fail(c, ZK_EINTERNAL, "device error word 0x%x", e);
}

}  // namespace zk

using namespace zk;

// Which XCDs do the workgroups of a launch land on?  (HW_REG_XCC_ID, MI355X_MICROARCH.md: workgroup dispatch.)
// The radix-sort pipeline hands runs of consecutive tiles to one XCD so that neighbouring output runs meet in
// one L2; that is only done when a probe launch sees exactly the ids 0..7, otherwise tiles go out in one order.
__global__ void xcd_probe_kernel(u32* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
static int probe_xcds(zk_ctx* c) {
    const int nb = 512;
    u32* d = nullptr;
    if (hipMalloc((void**)&d, nb * sizeof(u32)) != hipSuccess) return 1;
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(nb), dim3(64), 0, c->stream, d);
    std::vector<u32> h(nb, 0);
    int n = 1;
    if (hipMemcpyAsync(h.data(), d, nb * sizeof(u32), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
        hipStreamSynchronize(c->stream) == hipSuccess) {
        u32 seen = 0;
        bool ok = true;
        for (u32 v : h) { if (v > 7) ok = false; else seen |= 1u << v; }
        if (ok && seen == 0xffu) n = 8;
    }
    (void)hipFree(d);
    return n;
}

static std::string g_create_error = "no context";

#define ZK_CREATE_STEP(call)                                                                      \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            char b__[256];                                                                        \
            snprintf(b__, sizeof b__, "zk_create: %s failed: %s", #call, hipGetErrorString(e__)); \
            g_create_error = b__;                                                                 \
            if (c) zk_destroy(c);                                                                 \
            return nullptr;                                                                       \
        }                                                                                         \
    } while (0)

extern "C" {

zk_ctx* zk_create(int device, uint64_t workspace_bytes) {
    zk_ctx* c = nullptr;
    int ndev = 0;
    ZK_CREATE_STEP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        char b[128];
        snprintf(b, sizeof b, "zk_create: device %d requested, %d visible", device, ndev);
        g_create_error = b;
        return nullptr;
    }
    ZK_CREATE_STEP(hipSetDevice(device));
    c = new zk_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    ZK_CREATE_STEP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    ZK_CREATE_STEP(hipMalloc((void**)&c->d_ticket, sizeof(u32)));
    ZK_CREATE_STEP(hipMalloc((void**)&c->d_err, sizeof(u32)));
    ZK_CREATE_STEP(hipMalloc((void**)&c->d_scalars, 64 * sizeof(u64)));
    ZK_CREATE_STEP(hipHostMalloc((void**)&c->h_scalars, 64 * sizeof(u64), hipHostMallocDefault));
    // On the context's OWN stream, then waited for: hipMemset runs on the null stream, asynchronously for device memory, and a
    // non-blocking stream does not order against it -- with other threads keeping the null stream busy the clear of the
    // ticket counter could land in the middle of this context's first kernel (duplicate tickets, no scanner, spin timeout).
    ZK_CREATE_STEP(hipMemsetAsync(c->d_ticket, 0, sizeof(u32), c->stream));
    ZK_CREATE_STEP(hipMalloc((void**)&c->d_xticket, 8 * 32 * sizeof(u32)));
    ZK_CREATE_STEP(hipMemsetAsync(c->d_xticket, 0, 8 * 32 * sizeof(u32), c->stream));
    ZK_CREATE_STEP(hipMemsetAsync(c->d_err, 0, sizeof(u32), c->stream));
    ZK_CREATE_STEP(hipMemsetAsync(c->d_scalars, 0, 64 * sizeof(u64), c->stream));
    ZK_CREATE_STEP(hipStreamSynchronize(c->stream));
    c->num_xcd = probe_xcds(c);
    if (workspace_bytes) {
        ZK_CREATE_STEP(hipMalloc((void**)&c->arena, workspace_bytes));
        c->arena_size = workspace_bytes;
    }
    return c;
}

void zk_destroy(zk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)zk_comm_destroy(c);
    ring_destroy(c);
    for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (c->arena) (void)hipFree(c->arena);
    if (c->aux) (void)hipFree(c->aux);
    if (c->status) (void)hipFree(c->status);
    if (c->part16) (void)hipFree(c->part16);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    if (c->d_xticket) (void)hipFree(c->d_xticket);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_scalars) (void)hipFree(c->d_scalars);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* zk_last_error(zk_ctx* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }

int zk_set_stream(zk_ctx* c, void* hip_stream) {
    if (!c) return ZK_EINVAL;
    enter(c);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else {
        ZK_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return ZK_OK;
}

void* zk_get_stream(zk_ctx* c) { return c ? (void*)c->stream : nullptr; }

int zk_sync(zk_ctx* c) {
    if (!c) return ZK_EINVAL;
    enter(c);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_reserve(zk_ctx* c, uint64_t workspace_bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (workspace_bytes <= c->arena_size) return ZK_OK;
    arena_reset(c);
    void* p;
    ZK_TRY(arena_alloc(c, workspace_bytes, &p));
    arena_reset(c);
    return ZK_OK;
}

int zk_release_workspace(zk_ctx* c) {
    if (!c) return ZK_EINVAL;
    enter(c);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    if (c->arena) { ZK_HIP(c, hipFree(c->arena)); c->arena = nullptr; c->arena_size = 0; c->arena_off = 0; }
    if (c->aux) { ZK_HIP(c, hipFree(c->aux)); c->aux = nullptr; c->aux_size = 0; }
    if (c->status) { ZK_HIP(c, hipFree(c->status)); c->status = nullptr; c->status_words = 0; c->epoch = 0; }
    if (c->part16) { ZK_HIP(c, hipFree(c->part16)); c->part16 = nullptr; c->part16_words = 0; }
    return ZK_OK;
}

int zk_mem_info(zk_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    size_t f = 0, t = 0;
    ZK_HIP(c, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return ZK_OK;
}

int zk_tune(zk_ctx* c, int what, int value) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (what == ZK_TUNE_SORT_VARIANT) { c->sort_variant = value; return ZK_OK; }
    if (what == ZK_TUNE_PAIRS_VARIANT) { c->pairs_variant = value; return ZK_OK; }
    if (what == ZK_TUNE_SHORT_SORT) { c->short_sort = value; return ZK_OK; }
    if (what == ZK_TUNE_SIDE_DIV) { c->side_div = value; return ZK_OK; }
    if (what == ZK_TUNE_WIDE_TILES) { c->wide_tiles = value ? 1 : 0; return ZK_OK; }
    if (what == ZK_TUNE_PACKED_PAIRS) { c->packed_pairs = value ? 1 : 0; return ZK_OK; }
    if (what == ZK_TUNE_EARLY_COLLAPSE) { c->early_collapse = value < 0 ? 0 : (value > 3 ? 3 : value); return ZK_OK; }
    if (what == ZK_TUNE_COMM_SELF_LOOP) { c->comm_self_loop = value ? 1 : 0; return ZK_OK; }
    if (what == ZK_TUNE_COMM_CHUNK) { c->comm_chunk_bytes = value > 0 ? (uint64_t)value : 0; return ZK_OK; }
    if (what == ZK_TUNE_STREAM_PASS) {
#ifdef ZK_PHASES
        c->stream_pass = value < 0 ? 0 : value;          // (bits 8 and up: the measurement modes of tools/p0_phases.py)
#else
        if (value < 0 || value > 3) return fail(c, ZK_EINVAL, "stream pass variant %d (0, 1 or 3)", value);
        c->stream_pass = value;
#endif
        return ZK_OK;
    }
    if (what == ZK_TUNE_TAG_WORDS) { c->tag_words = value < 0 ? 0 : (value > 2 ? 2 : value); return ZK_OK; }
    if (what == ZK_TUNE_DEDUPE_VARIANT) { c->dedupe_variant = value < 0 ? -1 : (value & 3); return ZK_OK; }
    if (what == ZK_TUNE_KWAY) { c->kway = value < 0 ? 0 : (value > 2 ? 2 : value); return ZK_OK; }
    if (what == ZK_TUNE_TILE_SORT) { c->tile_sort = value ? 1 : 0; return ZK_OK; }
    if (what == ZK_TUNE_TAG_PASS) { c->tag_pass = value ? 1 : 0; return ZK_OK; }
    if (what == ZK_TUNE_DEDUPE_BITS) { c->dedupe_bits = value < 0 ? 0 : value; return ZK_OK; }
    if (what == ZK_TUNE_DEDUPE_LIMIT) { c->dedupe_limit = value < 1 ? 1 : (value > 65536 ? 65536 : value); return ZK_OK; }
    if (what == ZK_TUNE_STREAM_RANGES) { c->stream_ranges = value < 0 ? 0 : (value > 4096 ? 4096 : value); return ZK_OK; }
    if (what == ZK_TUNE_XCD_GROUP) {
        if (value < 0 || value > 32 || (value & (value - 1))) return fail(c, ZK_EINVAL, "xcd group must be 0 or a power of two <= 32");
        c->xcd_group = value;
        return ZK_OK;
    }
    return fail(c, ZK_EINVAL, "unknown tuning knob %d", what);
}

// diagnostic builds only (-DZK_STAMPS): where pass_kernel writes its per-tile time stamps
int zk_debug_buffer(zk_ctx* c, void* d_buf) {
    if (!c) return ZK_EINVAL;
    enter(c);
    c->dbg = (u64*)d_buf;
    return ZK_OK;
}

int zk_profile(zk_ctx* c, int enable) {
    if (!c) return ZK_EINVAL;
    enter(c);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    c->prof.clear();
    c->profile = enable != 0;
    return ZK_OK;
}

int zk_profile_read(zk_ctx* c, int tag, uint64_t* launches, double* total_ms, uint64_t* total_bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    uint64_t n = 0, bytes = 0;
    double ms = 0;
    for (auto& r : c->prof) {
        if (r.tag != tag) continue;
        float t = 0;
        ZK_HIP(c, hipEventElapsedTime(&t, r.a, r.b));
        n++; ms += t; bytes += r.bytes;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = bytes;
    return ZK_OK;
}

int zk_alloc(zk_ctx* c, uint64_t bytes, void** dptr) {
    if (!c || !dptr) return ZK_EINVAL;
    enter(c);
    *dptr = nullptr;
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return fail(c, ZK_ENOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return ZK_OK;
}

int zk_free(zk_ctx* c, void* dptr) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (dptr) { ZK_HIP(c, hipStreamSynchronize(c->stream)); ZK_HIP(c, hipFree(dptr)); }
    return ZK_OK;
}

int zk_host_alloc(zk_ctx* c, uint64_t bytes, void** ptr) {
    if (!c || !ptr) return ZK_EINVAL;
    enter(c);
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 256, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, ZK_ENOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return ZK_OK;
}

int zk_host_free(zk_ctx* c, void* ptr) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (ptr) { ZK_HIP(c, hipStreamSynchronize(c->stream)); ZK_HIP(c, hipHostFree(ptr)); }
    return ZK_OK;
}

int zk_upload_async(zk_ctx* c, void* dst_dev, const void* src_host, uint64_t bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    ZK_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    return ZK_OK;
}

int zk_upload(zk_ctx* c, void* dst_dev, const void* src_host, uint64_t bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    ZK_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_download(zk_ctx* c, void* dst_host, const void* src_dev, uint64_t bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    ZK_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_copy(zk_ctx* c, void* dst_dev, const void* src_dev, uint64_t bytes) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    ZK_HIP(c, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
    return ZK_OK;
}

}  // extern "C"
