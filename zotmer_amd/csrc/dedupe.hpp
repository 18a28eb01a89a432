// dedupe.hpp -- what the two block-dedupe kernels share (radix_sort.hip::dedupe_kernel, dedupe2.hip::dedupe2_kernel).
//
// K4 of DESIGN.md: replaces the run-length half of kmerize.merge (zotmer/commands/kmerize.py:41-132) -- the keys of one block (equal
// top bits, every copy of a k-mer inside) are counted in an LDS hash table and leave the kernel sorted, as words (key << pack | count).
#pragma once
#include "internal.hpp"

namespace zk {

struct DedupeArgs {
    const u64* kin;
    const u32* tin;     // or: the keys' low 32 bits only (TAGIN; the bits above are the block's number)
    u64 n;
    const u64* cuts;    // [chunks + 1]: chunk v = the block of keys whose top bits are v
    u64* out;           // block v writes its words from out + cuts[v] on; dedupe_unpack_kernel closes the gaps
    u64* nwords;        // [chunks] words of block v
    int tag_bits;       // key bits below the block bits
    int pack;
    u32* flags;         // |= 1: a table filled up, |= 2: some count went to the side list
    u32* counter;       // the next block to take
    u64* big;           // (key, count) pairs whose count does not fit `pack` bits
    u32* n_big;
    u32 big_cap;
    u32 chunks;         // tickets to hand out: blocks 0 .. chunks - 1, or (list) the blocks list[0 .. chunks - 1]
    u32* sub;           // or null: [chunks][64] entries of block v whose tag starts with the 6 bits j (the mirror sort groups by them)
    u32* bad;           // [bad_cap] blocks whose table filled up: they write nothing here, the host counts them by sorting
    u32* n_bad;
    u32 bad_cap;
    const u32* list;    // or null: the blocks to count (dedupe_kernel as the second chance of the blocks dedupe2_kernel declined)
    u32* retry;         // dedupe2_kernel: the blocks it declines (65 536 keys or more: 16-bit counts; a table that filled up) ...
    u32* n_retry;       // ... and how many; dedupe_kernel counts them afterwards, in its larger table with 32-bit counts
    u32 limit;          // ... blocks of this many keys or more are declined (<= 65 536)
    u64* dbg;           // or null (zk_debug_buffer + 8192 words): [workgroup][16] ticks per phase of a block, summed (tools/p0_phases.py)
};

#ifdef ZK_PHASES          // make CXXFLAGS_EXTRA=-DZK_PHASES: the diagnostic build tools/p0_phases.py reads
#define DD_PHASE(k) do { if (a.dbg) { const u32 now__ = (u32)__builtin_amdgcn_s_memtime(); ph[k] += now__ - tlast; tlast = now__; } } while (0)
#else
#define DD_PHASE(k) do { } while (0)
#endif

// dedupe2.hip: two workgroups per CU, 32-bit tags, 16-bit counts.  variant: zk_tune(ZK_TUNE_DEDUPE_VARIANT)
int launch_dedupe2(zk_ctx* c, const DedupeArgs& a, bool tagin, int variant);

}  // namespace zk
