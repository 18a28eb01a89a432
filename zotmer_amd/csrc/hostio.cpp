// hostio.cpp -- host-side (CPU) halves of the drop-in that feed and drain the GPU path:
//   * the on-disk vector codec of the sorted k-mer-set format: codec64 (zotmer/library/codec64.py)
//     and the delta transform around it (zotmer/library/files.py:85-110), byte-exact with the
//     streams the reference writes;
//   * FASTQ / FASTA text -> base stream (zotmer/library/file.py:19-52), the layout the encode
//     kernels read.
// These are product code (the CLI needs them to read and write real files); the test oracle has its
// own independent restatement in oracle/zk_oracle.c.  No GPU is touched here.
#include <stdint.h>
#include <string.h>

#include "../../include/zotk.h"

namespace {

inline int bit_length(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }

// width in bits of the fields of a word with tag n, as the reference's decoder table gives it
// (codec64.py:28-31: _codes[60 // i] = i, last assignment wins); 0 = no such tag
const int kDecodeWidth[16] = {0, 60, 30, 20, 15, 12, 10, 8, 7, 0, 6, 0, 5, 0, 0, 4};

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

}  // namespace

extern "C" {

// codec64.encode (codec64.py:82-120; same words as the streaming encoder :42-80).  Greedy: a word
// holds the longest run of n <= 6 values whose widest member fits 60/n bits.
// prev: when delta != 0 the values are ascending k-mers and each is replaced by its difference to
// the one before (first from 0) on the fly (files.delta, files.py:85-90).
int zk_codec64_encode(const uint64_t* vals, uint64_t n, int delta, uint64_t* words, uint64_t cap, uint64_t* n_words) {
    if (!n_words || (n && (!vals || !words))) return ZK_EINVAL;
    uint64_t nw = 0, i = 0, prev = 0;
    uint64_t v[6];
    while (i < n) {
        int cnt = 0, mw = 0;
        uint64_t p = prev;
        while (cnt < 6 && i + cnt < n) {
            const uint64_t x = delta ? vals[i + cnt] - p : vals[i + cnt];
            const int w = bit_length(x);
            const int mwx = w > mw ? w : mw;
            if (mwx > 60 / (cnt + 1)) break;
            v[cnt] = x;
            p = vals[i + cnt];
            mw = mwx;
            cnt++;
        }
        if (cnt == 0) return ZK_ERANGE;    // a value (or delta) >= 2^60 has no code
        if (nw >= cap) return ZK_ENOSPC;
        const int b = 60 / cnt;
        uint64_t w = 0;
        for (int m = cnt - 1; m >= 0; m--) w = (w << b) | v[m];
        words[nw++] = (w << 4) | (uint64_t)cnt;
        prev = vals[i + cnt - 1];
        i += cnt;
    }
    *n_words = nw;
    return ZK_OK;
}

// number of values a word stream holds (sum of the tags)
int zk_codec64_count(const uint64_t* words, uint64_t nw, uint64_t* n_values) {
    if (!n_values || (nw && !words)) return ZK_EINVAL;
    uint64_t n = 0;
    for (uint64_t j = 0; j < nw; j++) {
        const int tag = (int)(words[j] & 15);
        if (!kDecodeWidth[tag]) return ZK_ERANGE;     // KeyError in the reference
        n += tag;
    }
    *n_values = n;
    return ZK_OK;
}

// codec64.decode / decodeList (codec64.py:122-151); delta != 0 also undoes the delta transform
// (files.undelta, files.py:100-104), i.e. yields the k-mers themselves.
int zk_codec64_decode(const uint64_t* words, uint64_t nw, int delta, uint64_t* out, uint64_t cap, uint64_t* n_out) {
    if (!n_out || (nw && (!words || !out))) return ZK_EINVAL;
    uint64_t n = 0, acc = 0;
    for (uint64_t j = 0; j < nw; j++) {
        uint64_t w = words[j];
        const int m0 = (int)(w & 15);
        const int b = kDecodeWidth[m0];
        if (!b) return ZK_ERANGE;
        if (n + m0 > cap) return ZK_ENOSPC;
        w >>= 4;
        const uint64_t msk = (1ull << b) - 1;
        for (int m = 0; m < m0; m++) {
            const uint64_t x = w & msk;
            w >>= b;
            if (delta) { acc += x; out[n++] = acc; } else out[n++] = x;
        }
    }
    *n_out = n;
    return ZK_OK;
}

// ---- text parsers ---------------------------------------------------------------------------------
// Both consume a chunk of a file and append to a base stream (every sequence followed by '\n').
// `state` carries what must survive between chunks; zero it before the first chunk.  A chunk must
// end at a line end unless final != 0 (feed whole lines; *consumed tells how much was used, the
// rest is to be presented again in front of the next chunk).

// file.readFastq (file.py:38-52): lines are taken in groups of four, each stripped; the second of a
// group is the sequence; a trailing group of fewer than four lines is dropped (:51-52 never fires).
// Only whole records are consumed, so nothing but the record count crosses chunks:
// state[1] = records so far.
int zk_parse_fastq(const char* buf, uint64_t len, int final, uint64_t state[4], uint8_t* out, uint64_t out_cap,
                   uint64_t* out_len, uint64_t* consumed) {
    if (!state || !out_len || !consumed || (len && !buf)) return ZK_EINVAL;
    uint64_t pos = 0, o = *out_len, records = state[1];
    int rc = ZK_OK;
    while (pos < len) {
        // the four lines of the next record: [ls[i], le[i])
        uint64_t ls[4], le[4], p = pos;
        int got = 0;
        while (got < 4 && p < len) {
            const char* nl = (const char*)memchr(buf + p, '\n', len - p);
            if (nl) { ls[got] = p; le[got] = (uint64_t)(nl - buf); p = le[got] + 1; got++; }
            else if (final) { ls[got] = p; le[got] = len; p = len; got++; }   // last line without '\n'
            else break;
        }
        if (got < 4) { if (final) pos = len; break; }      // partial record: wait for more, or drop at the end
        uint64_t a = ls[1], b = le[1];                      // strip
        while (a < b && is_space((unsigned char)buf[a])) a++;
        while (b > a && is_space((unsigned char)buf[b - 1])) b--;
        if (o + (b - a) + 1 > out_cap) { rc = ZK_ENOSPC; break; }
        memcpy(out + o, buf + a, b - a);
        out[o + (b - a)] = '\n';
        o += (b - a) + 1;
        records++;
        pos = p;
    }
    state[1] = records;
    *out_len = o;
    *consumed = pos;
    return rc;
}

// file.readFasta (file.py:19-36): a line starting with '>' (after strip) opens a record; the stripped
// lines up to the next header are concatenated; text before the first header is ignored.
// state[0] = 1 once inside a record, state[1] = records so far.
int zk_parse_fasta(const char* buf, uint64_t len, int final, uint64_t state[4], uint8_t* out, uint64_t out_cap,
                   uint64_t* out_len, uint64_t* consumed) {
    if (!state || !out_len || !consumed || (len && !buf)) return ZK_EINVAL;
    uint64_t pos = 0, o = *out_len;
    uint64_t in_rec = state[0], records = state[1];
    while (pos < len) {
        const char* nl = (const char*)memchr(buf + pos, '\n', len - pos);
        uint64_t end;
        if (nl) end = (uint64_t)(nl - buf);
        else if (final) end = len;
        else break;
        uint64_t a = pos, b = end;
        while (a < b && is_space((unsigned char)buf[a])) a++;
        while (b > a && is_space((unsigned char)buf[b - 1])) b--;
        if (b > a && buf[a] == '>') {
            if (o + 1 > out_cap) { state[0] = in_rec; state[1] = records; *out_len = o; *consumed = pos; return ZK_ENOSPC; }
            if (in_rec) out[o++] = '\n';               // close the previous record
            in_rec = 1;
            records++;
        } else if (in_rec) {
            if (o + (b - a) + 1 > out_cap) { state[0] = in_rec; state[1] = records; *out_len = o; *consumed = pos; return ZK_ENOSPC; }
            memcpy(out + o, buf + a, b - a);
            o += b - a;
        }
        pos = nl ? end + 1 : end;
    }
    if (final && in_rec) {
        if (o + 1 > out_cap) { state[0] = in_rec; state[1] = records; *out_len = o; *consumed = pos; return ZK_ENOSPC; }
        out[o++] = '\n';
        in_rec = 0;
    }
    state[0] = in_rec; state[1] = records;
    *out_len = o;
    *consumed = pos;
    return ZK_OK;
}

}  // extern "C"
