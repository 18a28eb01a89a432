// ingest.hip -- f2 of SURVEY.md section 8(f): file bytes <-> device memory without the host ever parsing them.
//
// Reference being replaced: file.readFastq / openFile (zotmer/library/file.py:38-52,79-123: four stripped lines per
// record, `gunzip -c` pipes for .gz) feeding reads.next (zotmer/library/reads.py:86-98) one Python string at a time.
//
// Here a FASTQ file is text that the GPU parses (zk_fastq_mask, codec.hip): the host only has to move bytes.
//   zk_source   a file (plain, or gzip detected by its magic) read AHEAD of the device: a background thread fills a
//               ring of page-locked buffers -- plain files by parallel pread() of slices, gzip streams by zlib inflate
//               (multi-member files included) -- and queues each filled buffer as one asynchronous H2D copy on a copy
//               stream of its own, so that reading / inflating, PCIe and the kernels of the previous batch overlap.
//               The caller names the device buffer of the next batch (zk_source_start), computes on the current
//               one, then waits (zk_source_finish).  Batches are cut at line ends on the DEVICE (zk_last_newline); the
//               few bytes after the cut are carried to the front of the next device buffer with a device copy.
//   zk_device_to_file / zk_file_to_device
//               the members of a k-mer set (codec64 word streams produced / consumed by codec.hip) between device
//               memory and a file region through the same ring, pwrite / pread in parallel slices.
// No device kernel here except the newline scan; everything else is threads, pread/pwrite, zlib and hipMemcpyAsync.
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace zk {

constexpr uint64_t RING_SLOT = 32ull << 20;      // bytes per page-locked slot
constexpr int RING_SLOTS = 4;

struct Ring {
    u8* slot[RING_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[RING_SLOTS];
    hipStream_t stream = nullptr;
    bool ok = false;
};

// one ring per context, made on first use, kept until zk_destroy (page-locking memory is slow: ~0.1 s per 100 MB)
static int ring_get(zk_ctx* c, Ring** out) {
    if (!c->ring) {
        Ring* r = new Ring();
        ZK_HIP(c, hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
        for (int i = 0; i < RING_SLOTS; i++) {
            ZK_HIP(c, hipHostMalloc((void**)&r->slot[i], RING_SLOT, hipHostMallocDefault));
            ZK_HIP(c, hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming));
        }
        r->ok = true;
        c->ring = r;
    }
    *out = (Ring*)c->ring;
    return ZK_OK;
}

void ring_destroy(zk_ctx* c) {
    Ring* r = (Ring*)c->ring;
    if (!r) return;
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (int i = 0; i < RING_SLOTS; i++) {
        if (r->slot[i]) { (void)hipHostFree(r->slot[i]); (void)hipEventDestroy(r->done[i]); }
    }
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
    c->ring = nullptr;
}

// [off, off + len) of fd into dst with `threads` parallel pread()s; returns bytes read (short only at end of file) or -1
static int64_t pread_parallel(int fd, u8* dst, uint64_t off, uint64_t len, int threads) {
    if (threads < 1) threads = 1;
    const uint64_t per = (len + threads - 1) / threads;
    std::vector<int64_t> got(threads, 0);
    auto work = [&](int t) {
        uint64_t a = (uint64_t)t * per, b = a + per < len ? a + per : len;
        int64_t n = 0;
        while (a < b) {
            const ssize_t r = pread(fd, dst + a, b - a, (off_t)(off + a));
            if (r < 0) { if (errno == EINTR) continue; n = -1; break; }
            if (r == 0) break;
            a += (uint64_t)r; n += r;
        }
        got[t] = n;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    int64_t total = 0;
    for (int t = 0; t < threads; t++) {          // slices are in file order: once the file ends every later slice reads nothing
        if (got[t] < 0) return -1;
        total += got[t];
    }
    return total;
}

static int pwrite_parallel(int fd, const u8* src, uint64_t off, uint64_t len, int threads) {
    if (threads < 1) threads = 1;
    const uint64_t per = (len + threads - 1) / threads;
    std::atomic<int> bad{0};
    auto work = [&](int t) {
        uint64_t a = (uint64_t)t * per, b = a + per < len ? a + per : len;
        while (a < b) {
            const ssize_t r = pwrite(fd, src + a, b - a, (off_t)(off + a));
            if (r < 0) { if (errno == EINTR) continue; bad = errno ? errno : EIO; return; }
            a += (uint64_t)r;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return bad.load();
}

}  // namespace zk

using namespace zk;

struct zk_source {
    zk_ctx* c = nullptr;
    int fd = -1;
    bool gz = false;
    uint64_t pos = 0;            // plain: next file offset; gz: compressed bytes consumed
    int threads = 4;
    bool eof = false;
    // gzip
    z_stream zs;
    bool zs_live = false;
    std::vector<u8> zin;
    uint64_t zin_have = 0, zin_off = 0;
    bool file_end = false;
    // one request in flight
    std::thread worker;
    bool busy = false;
    uint64_t got = 0;
    int rc = ZK_OK;
    std::string err;
};

namespace zk {

// fill `slot` (cap bytes) with the next decompressed bytes; returns the count, 0 at the end of the stream, -1 on error
static int64_t gz_fill(zk_source* s, u8* slot, uint64_t cap) {
    z_stream& z = s->zs;
    z.next_out = slot;
    uint64_t produced = 0;
    while (produced < cap) {
        if (s->zin_off == s->zin_have && !s->file_end) {
            const ssize_t r = read(s->fd, s->zin.data(), s->zin.size());
            if (r < 0) { if (errno == EINTR) continue; s->err = std::string("read: ") + strerror(errno); return -1; }
            if (r == 0) s->file_end = true;
            s->zin_have = (uint64_t)(r > 0 ? r : 0); s->zin_off = 0;
        }
        if (s->zin_off == s->zin_have && s->file_end) {
            if (s->zs_live) { s->err = "gzip stream is truncated"; return -1; }
            break;
        }
        if (!s->zs_live) {                            // (next) member
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, 15 + 32) != Z_OK) { s->err = "inflateInit2 failed"; return -1; }
            s->zs_live = true;
        }
        z.next_in = s->zin.data() + s->zin_off;
        z.avail_in = (uInt)(s->zin_have - s->zin_off);
        const uint64_t room = cap - produced;
        z.next_out = slot + produced;
        z.avail_out = (uInt)(room > (1u << 30) ? (1u << 30) : room);
        const uInt out0 = z.avail_out, in0 = z.avail_in;
        const int r = inflate(&z, Z_NO_FLUSH);
        produced += out0 - z.avail_out;
        s->zin_off += in0 - z.avail_in;
        if (r == Z_STREAM_END) { inflateEnd(&z); s->zs_live = false; continue; }      // another member may follow
        if (r != Z_OK && r != Z_BUF_ERROR) { s->err = std::string("inflate: ") + (z.msg ? z.msg : "data error"); return -1; }
        if (r == Z_BUF_ERROR && out0 == z.avail_out && in0 == z.avail_in && s->file_end) { s->err = "gzip stream is truncated"; return -1; }
    }
    return (int64_t)produced;
}

// the body of one request: up to `cap` bytes of the source into d_dst, through the ring
static void source_run(zk_source* s, u8* d_dst, uint64_t cap) {
    zk_ctx* c = s->c;
    (void)hipSetDevice(c->device);
    Ring* r = (Ring*)c->ring;
    uint64_t done = 0;
    int k = 0;
    s->rc = ZK_OK;
    while (done < cap && !s->eof) {
        const int i = k % RING_SLOTS;
        if (k >= RING_SLOTS && hipEventSynchronize(r->done[i]) != hipSuccess) { s->rc = ZK_EHIP; s->err = "hipEventSynchronize failed"; break; }
        const uint64_t want = cap - done < RING_SLOT ? cap - done : RING_SLOT;
        int64_t n;
        if (s->gz) n = gz_fill(s, r->slot[i], want);
        else {
            n = pread_parallel(s->fd, r->slot[i], s->pos, want, s->threads);
            if (n < 0) s->err = std::string("pread: ") + strerror(errno);
            else s->pos += (uint64_t)n;
        }
        if (n < 0) { s->rc = ZK_EINVAL; break; }
        if ((uint64_t)n < want) s->eof = true;
        if (n > 0) {
            if (hipMemcpyAsync(d_dst + done, r->slot[i], (size_t)n, hipMemcpyHostToDevice, r->stream) != hipSuccess ||
                hipEventRecord(r->done[i], r->stream) != hipSuccess) { s->rc = ZK_EHIP; s->err = "hipMemcpyAsync (H2D) failed"; break; }
            done += (uint64_t)n;
            k++;
        }
    }
    if (hipStreamSynchronize(r->stream) != hipSuccess && s->rc == ZK_OK) { s->rc = ZK_EHIP; s->err = "copy stream failed"; }
    s->got = done;
}

// position just after the last '\n' of text[0, n) (0 when there is none): where a batch of whole lines ends
__global__ void last_newline_kernel(const u8* __restrict__ text, u64 n, u64* out) {
    // the answer is almost always within the last few hundred bytes: scan backwards in steps of the block
    __shared__ u64 best;
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    for (u64 end = n; end > 0;) {
        const u64 begin = end > blockDim.x ? end - blockDim.x : 0;
        const u64 i = begin + threadIdx.x;
        if (i < end && text[i] == '\n') atomicMax(&best, i + 1);
        __syncthreads();
        if (best) break;
        end = begin;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = best;
}

}  // namespace zk

extern "C" {

zk_source* zk_source_open(zk_ctx* c, const char* path, int threads) {
    if (!c || !path) return nullptr;
    enter(c);
    Ring* r;
    if (ring_get(c, &r) != ZK_OK) return nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { fail(c, ZK_EINVAL, "cannot open %s: %s", path, strerror(errno)); return nullptr; }
    zk_source* s = new zk_source();
    s->c = c; s->fd = fd;
    s->threads = threads > 0 ? (threads > 32 ? 32 : threads) : 4;
    unsigned char magic[2] = {0, 0};
    if (pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        s->gz = true;
        s->zin.resize(4u << 20);
    }
    return s;
}

int zk_source_is_gzip(zk_source* s) { return s && s->gz ? 1 : 0; }

int zk_source_start(zk_source* s, uint8_t* d_dst, uint64_t cap) {
    if (!s) return ZK_EINVAL;
    zk_ctx* c = s->c;
    enter(c);
    if (s->busy) return fail(c, ZK_EINVAL, "zk_source_start: a request is already in flight");
    s->got = 0;
    if (cap == 0 || s->eof) { s->rc = ZK_OK; s->busy = true; return ZK_OK; }
    if (!d_dst) return fail(c, ZK_EINVAL, "zk_source_start: no destination");
    s->busy = true;
    s->worker = std::thread(source_run, s, (u8*)d_dst, cap);
    return ZK_OK;
}

int zk_source_finish(zk_source* s, uint64_t* n_bytes, int* eof) {
    if (!s) return ZK_EINVAL;
    zk_ctx* c = s->c;
    if (!s->busy) return fail(c, ZK_EINVAL, "zk_source_finish: nothing in flight");
    if (s->worker.joinable()) s->worker.join();
    s->busy = false;
    if (n_bytes) *n_bytes = s->got;
    if (eof) *eof = s->eof ? 1 : 0;
    if (s->rc != ZK_OK) return fail(c, s->rc, "source: %s", s->err.c_str());
    return ZK_OK;
}

void zk_source_close(zk_source* s) {
    if (!s) return;
    if (s->worker.joinable()) s->worker.join();
    if (s->zs_live) inflateEnd(&s->zs);
    if (s->fd >= 0) close(s->fd);
    delete s;
}

int zk_last_newline(zk_ctx* c, const uint8_t* d_text, uint64_t n, uint64_t* cut) {
    if (!c || !cut) return ZK_EINVAL;
    enter(c);
    *cut = 0;
    if (n == 0) return ZK_OK;
    u64* d = c->d_scalars + 19;
    hipLaunchKernelGGL(last_newline_kernel, dim3(1), dim3(1024), 0, c->stream, (const u8*)d_text, (u64)n, d);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->h_scalars + 19, d, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *cut = c->h_scalars[19];
    return ZK_OK;
}

int zk_device_to_file(zk_ctx* c, const void* d_src, uint64_t bytes, int fd, uint64_t file_offset, int threads) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    if (!d_src || fd < 0) return fail(c, ZK_EINVAL, "zk_device_to_file: bad argument");
    Ring* r;
    ZK_TRY(ring_get(c, &r));
    // the file's blocks in one request instead of one extension per pwrite slice (where the file system can: errors are not errors here)
    if (!getenv("ZOT_NO_FALLOCATE")) (void)fallocate(fd, 0, (off_t)file_offset, (off_t)bytes);
    ZK_HIP(c, hipStreamSynchronize(c->stream));              // the producer of d_src ran on the compute stream
    const uint64_t chunks = div_up(bytes, RING_SLOT);
    auto d2h = [&](uint64_t k) -> hipError_t {
        const int i = (int)(k % RING_SLOTS);
        const uint64_t off = k * RING_SLOT, len = bytes - off < RING_SLOT ? bytes - off : RING_SLOT;
        hipError_t e = hipMemcpyAsync(r->slot[i], (const char*)d_src + off, len, hipMemcpyDeviceToHost, r->stream);
        return e != hipSuccess ? e : hipEventRecord(r->done[i], r->stream);
    };
    // the D2H copy of chunk k + 1 runs while chunk k is written (pwrite in parallel slices on this thread's helpers);
    // a slot is reused RING_SLOTS chunks later, long after its write has returned
    ZK_HIP(c, d2h(0));
    for (uint64_t k = 0; k < chunks; k++) {
        if (k + 1 < chunks) ZK_HIP(c, d2h(k + 1));
        const int i = (int)(k % RING_SLOTS);
        const uint64_t off = k * RING_SLOT, len = bytes - off < RING_SLOT ? bytes - off : RING_SLOT;
        ZK_HIP(c, hipEventSynchronize(r->done[i]));
        const int e = pwrite_parallel(fd, r->slot[i], file_offset + off, len, threads);
        if (e) { (void)hipStreamSynchronize(r->stream); return fail(c, ZK_EINVAL, "write failed: %s", strerror(e)); }
    }
    return ZK_OK;
}

int zk_file_to_device(zk_ctx* c, int fd, uint64_t file_offset, uint64_t bytes, void* d_dst, int threads) {
    if (!c) return ZK_EINVAL;
    enter(c);
    if (bytes == 0) return ZK_OK;
    if (!d_dst || fd < 0) return fail(c, ZK_EINVAL, "zk_file_to_device: bad argument");
    Ring* r;
    ZK_TRY(ring_get(c, &r));
    uint64_t done = 0;
    for (uint64_t k = 0; done < bytes; k++) {
        const int i = (int)(k % RING_SLOTS);
        if (k >= RING_SLOTS) ZK_HIP(c, hipEventSynchronize(r->done[i]));
        const uint64_t len = bytes - done < RING_SLOT ? bytes - done : RING_SLOT;
        const int64_t n = pread_parallel(fd, r->slot[i], file_offset + done, len, threads);
        if (n < 0 || (uint64_t)n < len) return fail(c, ZK_EINVAL, "read failed or file too short (%lld of %llu bytes)", (long long)n, (unsigned long long)len);
        ZK_HIP(c, hipMemcpyAsync((char*)d_dst + done, r->slot[i], len, hipMemcpyHostToDevice, r->stream));
        ZK_HIP(c, hipEventRecord(r->done[i], r->stream));
        done += len;
    }
    ZK_HIP(c, hipStreamSynchronize(r->stream));
    return ZK_OK;
}

}  // extern "C"
