// hostio_san_driver.cpp -- drives the product's host code (zotmer_amd/csrc/hostio.cpp: codec64 + delta, FASTQ / FASTA chunk
// parsers, standing in for zotmer/library/codec64.py:42-151, library/files.py:85-110, library/file.py:19-52) under
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU.  Every buffer handed to the library is a heap block of EXACTLY the
// size the call is told about, so any read or write past it stops the run.  tests/test_hostio_sanitizers.py feeds it the golden
// streams and compares what comes back; `fuzz` walks random and hostile inputs on its own.
//   driver enc <delta> <values.bin> <words.bin>     driver dec <delta> <words.bin> <values.bin>
//   driver fastq|fasta <chunk> <text> <stream.bin>  driver fuzz <seed> <rounds>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/zotk.h"

static std::vector<unsigned char> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    std::vector<unsigned char> v;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}
static void spill(const char* path, const void* p, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, n, f) != n) { perror(path); exit(2); }
    fclose(f);
}
// exact-size heap copies (malloc(0) is avoided: one byte that is never offered to the callee)
template <class T> static T* exact(const T* src, size_t n) {
    T* p = (T*)malloc(n ? n * sizeof(T) : 1);
    if (n && src) memcpy(p, src, n * sizeof(T));
    return p;
}

static int parse(bool fasta, const unsigned char* text, size_t len, size_t chunk, std::vector<unsigned char>& out, uint64_t* records) {
    uint64_t state[4] = {0, 0, 0, 0};
    std::string carry;
    size_t pos = 0;
    uint64_t out_len = 0;
    uint64_t cap = len + 2;          // a base stream is never longer than its text + the closing newline
    unsigned char* o = (unsigned char*)malloc(cap);
    do {
        const size_t take = len - pos < chunk ? len - pos : chunk;
        carry.append((const char*)text + pos, take);
        pos += take;
        const int final = pos == len;
        char* buf = exact(carry.data(), carry.size());
        uint64_t consumed = 0;
        const int rc = (fasta ? zk_parse_fasta : zk_parse_fastq)(buf, carry.size(), final, state, o, cap, &out_len, &consumed);
        free(buf);
        if (rc != ZK_OK) { free(o); return rc; }
        if (consumed > carry.size()) { fprintf(stderr, "consumed %llu of %zu\n", (unsigned long long)consumed, carry.size()); exit(3); }
        carry.erase(0, consumed);
    } while (pos < len);
    out.assign(o, o + out_len);
    free(o);
    *records = state[1];
    return ZK_OK;
}

static uint64_t rng_state;
static uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static void fuzz(uint64_t seed, int rounds) {
    rng_state = seed;
    for (int r = 0; r < rounds; r++) {
        // codec64: values of every width, runs of equal widths, the 2^60 edge
        const size_t n = rnd() % 300;
        std::vector<uint64_t> v(n);
        bool too_big = false;
        for (size_t i = 0; i < n; i++) {
            const int w = (int)(rnd() % 62);
            v[i] = w ? (rnd() >> (64 - w)) : 0;
            if (r % 7 == 0 && rnd() % 40 == 0) v[i] = 1ull << 60;
            too_big |= v[i] >= (1ull << 60);
        }
        uint64_t* vin = exact(v.data(), n);
        uint64_t* words = exact((uint64_t*)nullptr, n);
        uint64_t nw = 0;
        int rc = zk_codec64_encode(vin, n, 0, words, n, &nw);
        if (too_big != (rc == ZK_ERANGE)) { fprintf(stderr, "fuzz %d: encode rc %d, too_big %d\n", r, rc, (int)too_big); exit(3); }
        if (rc == ZK_OK) {
            uint64_t cnt = 0, got = 0;
            uint64_t* w2 = exact(words, nw);
            if (zk_codec64_count(w2, nw, &cnt) != ZK_OK || cnt != n) { fprintf(stderr, "fuzz %d: count %llu of %zu\n", r, (unsigned long long)cnt, n); exit(3); }
            uint64_t* back = exact((uint64_t*)nullptr, n);
            if (zk_codec64_decode(w2, nw, 0, back, n, &got) != ZK_OK || got != n || (n && memcmp(back, vin, 8 * n))) { fprintf(stderr, "fuzz %d: round trip\n", r); exit(3); }
            // one entry short: must say so, not write
            if (n) { uint64_t* small = exact((uint64_t*)nullptr, n - 1); if (zk_codec64_decode(w2, nw, 0, small, n - 1, &got) != ZK_ENOSPC) { fprintf(stderr, "fuzz %d: short buffer\n", r); exit(3); } free(small); }
            if (nw) { uint64_t* fewer = exact((uint64_t*)nullptr, nw - 1); if (zk_codec64_encode(vin, n, 0, fewer, nw - 1, &got) != ZK_ENOSPC) { fprintf(stderr, "fuzz %d: short words\n", r); exit(3); } free(fewer); }
            free(back); free(w2);
        }
        // delta: ascending k-mers
        std::vector<uint64_t> k(n);
        uint64_t acc = 0;
        for (size_t i = 0; i < n; i++) { acc += 1 + (rnd() >> (4 + rnd() % 56)); k[i] = acc & ((1ull << 62) - 1); if (i && k[i] <= k[i - 1]) k[i] = k[i - 1] + 1; }
        uint64_t* kin = exact(k.data(), n);
        rc = zk_codec64_encode(kin, n, 1, words, n, &nw);
        if (rc == ZK_OK) {
            uint64_t got = 0;
            uint64_t* back = exact((uint64_t*)nullptr, n);
            uint64_t* w2 = exact(words, nw);
            if (zk_codec64_decode(w2, nw, 1, back, n, &got) != ZK_OK || got != n || (n && memcmp(back, kin, 8 * n))) { fprintf(stderr, "fuzz %d: delta round trip\n", r); exit(3); }
            free(back); free(w2);
        } else if (rc != ZK_ERANGE) { fprintf(stderr, "fuzz %d: delta encode rc %d\n", r, rc); exit(3); }
        // hostile word streams: every tag, random payloads
        const size_t m = rnd() % 64;
        std::vector<uint64_t> junk(m);
        for (auto& x : junk) x = rnd();
        uint64_t* jw = exact(junk.data(), m);
        uint64_t cnt = 0, got = 0;
        if (zk_codec64_count(jw, m, &cnt) == ZK_OK) {
            uint64_t* back = exact((uint64_t*)nullptr, cnt);
            if (zk_codec64_decode(jw, m, (int)(r & 1), back, cnt, &got) != ZK_OK || got != cnt) { fprintf(stderr, "fuzz %d: junk decode\n", r); exit(3); }
            free(back);
        }
        free(jw); free(kin); free(words); free(vin);
        // parsers: random text with every kind of line end, fed in random chunk sizes; the result must not depend on the chunks
        const size_t tl = rnd() % 3000;
        std::vector<unsigned char> text(tl);
        const char alphabet[] = "ACGTNacgt>@+ \t\r\n\n\n\nIIII";
        for (auto& c : text) c = (unsigned char)alphabet[rnd() % (sizeof alphabet - 1)];
        for (int fasta = 0; fasta < 2; fasta++) {
            std::vector<unsigned char> whole, pieces;
            uint64_t r1 = 0, r2 = 0;
            const int a = parse(fasta, text.data(), tl, tl + 1, whole, &r1);
            const int b = parse(fasta, text.data(), tl, 1 + rnd() % 97, pieces, &r2);
            if (a != ZK_OK || b != ZK_OK || whole != pieces || r1 != r2) { fprintf(stderr, "fuzz %d: parser %d depends on the chunks (%d %d)\n", r, fasta, a, b); exit(3); }
        }
    }
    printf("fuzz ok: %d rounds\n", rounds);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "fuzz" && argc == 4) { fuzz(strtoull(argv[2], nullptr, 10), atoi(argv[3])); return 0; }
    if ((cmd == "enc" || cmd == "dec") && argc == 5) {
        const int delta = atoi(argv[2]);
        std::vector<unsigned char> in = slurp(argv[3]);
        const size_t n = in.size() / 8;
        uint64_t* src = exact((const uint64_t*)in.data(), n);
        int rc;
        if (cmd == "enc") {
            uint64_t* words = exact((uint64_t*)nullptr, n);
            uint64_t nw = 0;
            rc = zk_codec64_encode(src, n, delta, words, n, &nw);
            if (rc == ZK_OK) spill(argv[4], words, 8 * nw);
            free(words);
        } else {
            uint64_t cnt = 0, got = 0;
            rc = zk_codec64_count(src, n, &cnt);
            if (rc == ZK_OK) {
                uint64_t* out = exact((uint64_t*)nullptr, cnt);
                rc = zk_codec64_decode(src, n, delta, out, cnt, &got);
                if (rc == ZK_OK && got != cnt) rc = ZK_EINTERNAL;
                if (rc == ZK_OK) spill(argv[4], out, 8 * cnt);
                free(out);
            }
        }
        free(src);
        printf("rc %d\n", rc);
        return 0;
    }
    if ((cmd == "fastq" || cmd == "fasta") && argc == 5) {
        std::vector<unsigned char> text = slurp(argv[3]), out;
        uint64_t records = 0;
        const int rc = parse(cmd == "fasta", text.data(), text.size(), (size_t)strtoull(argv[2], nullptr, 10), out, &records);
        if (rc != ZK_OK) { printf("rc %d\n", rc); return 0; }
        spill(argv[4], out.data(), out.size());
        printf("rc 0 records %llu\n", (unsigned long long)records);
        return 0;
    }
    return 2;
}
