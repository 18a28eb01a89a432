"""
Counter-based synthetic read generator.

Every base is a pure function of (seed, read index, position), built on the splitmix64
finaliser, so the numpy version here and the HIP version in csrc/synth.hip produce the
same bytes without sharing any state; tests check the two against each other.

  rnd(seed, tag, i) = mix64(mix64(seed + tag) + i)          (all arithmetic mod 2**64)

Genome-sampled mode: the genome is never stored -- base g of the genome is rnd(seed,1,g) & 3.
Read i starts at rnd(seed,2,i) % (G - L + 1) on strand rnd(seed,3,i) & 1; base j gets a
substitution when the low 32 bits of rnd(seed,4,i*L+j) fall below sub_thr and becomes 'N'
when the high 32 bits fall below n_thr (thresholds are fractions of 2**32).
Uniform mode: base = rnd(seed,6,i*L+j) & 3, with the same 'N' rule.

The device input format ("base stream") is every read followed by one '\n'.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

DEFAULT_SEED = 20261004


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def rnd(seed, tag, i):
    with np.errstate(over="ignore"):
        base = mix64(np.uint64((int(seed) + int(tag)) & 0xFFFFFFFFFFFFFFFF))
        return mix64(base + np.asarray(i, dtype=np.uint64))


def frac32(p):
    """A probability as a threshold on a uniform 32-bit draw."""
    return int(round(p * 4294967296.0))


def reads_matrix(seed, first, count, L, genome=0, sub_thr=0, n_thr=0):
    """uint8[count, L] of ASCII bases for reads first .. first+count-1.
    genome == 0 selects uniform mode; otherwise it is the genome length G (>= L)."""
    i = np.arange(first, first + count, dtype=np.uint64)[:, None]
    j = np.arange(L, dtype=np.uint64)[None, :]
    idx = i * np.uint64(L) + j
    if genome:
        start = rnd(seed, 2, i) % np.uint64(genome - L + 1)
        strand = rnd(seed, 3, i) & np.uint64(1)
        gpos = np.where(strand == 1, start + np.uint64(L - 1) - j, start + j)
        b = rnd(seed, 1, gpos) & np.uint64(3)
        b = np.where(strand == 1, np.uint64(3) - b, b)
        e = rnd(seed, 4, idx)
        if sub_thr:
            hit = (e & np.uint64(0xFFFFFFFF)) < np.uint64(sub_thr)
            alt = (b + np.uint64(1) + rnd(seed, 5, idx) % np.uint64(3)) & np.uint64(3)
            b = np.where(hit, alt, b)
    else:
        b = rnd(seed, 6, idx) & np.uint64(3)
        e = rnd(seed, 4, idx)
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[b.astype(np.intp)]
    if n_thr:
        out = np.where((e >> np.uint64(32)) < np.uint64(n_thr), np.uint8(ord("N")), out)
    return np.ascontiguousarray(out, dtype=np.uint8)


def base_stream(seed, first, count, L, **kw):
    """The device input: uint8[count*(L+1)], each read followed by '\\n'."""
    m = reads_matrix(seed, first, count, L, **kw)
    s = np.full((count, L + 1), ord("\n"), dtype=np.uint8)
    s[:, :L] = m
    return s.reshape(-1)


def read_strings(seed, first, count, L, **kw):
    m = reads_matrix(seed, first, count, L, **kw)
    return [bytes(r).decode() for r in m]


def fastq_text(seed, first, count, L, **kw):
    """FASTQ text '@r<i>\\n<seq>\\n+\\n<I*L>\\n' for the same reads."""
    q = "I" * L
    return "".join("@r%d\n%s\n+\n%s\n" % (first + k, s, q)
                   for k, s in enumerate(read_strings(seed, first, count, L, **kw)))


# ---- synthetic sorted k-mer sets (BASELINE configs 3 and 4); the HIP version is csrc/partition.hip ----------

def set_keys_raw(seed, first, count, key_bits, mul=1, add=0, mod=1 << 62):
    """Element first+i of an affine walk through a pool of `mod` random keys: rnd(seed, 7, (mul*(first+i)+add) % mod)
    masked to key_bits -- distinct pool indices while first+i < mod and gcd(mul, mod) == 1 (a draw without
    replacement).  Unsorted, may hold the odd duplicate value."""
    i = np.arange(first, first + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        j = (np.uint64(mul) * i + np.uint64(add)) % np.uint64(mod)
    mask = np.uint64((1 << key_bits) - 1) if key_bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    return rnd(seed, 7, j) & mask


def set_keys(seed, first, count, key_bits, **kw):
    """The sorted distinct keys of set_keys_raw."""
    return np.unique(set_keys_raw(seed, first, count, key_bits, **kw))


def set_counts(seed, keys):
    """Geometric counts (mean 8) in integer arithmetic: successive 3-bit groups of rnd(seed, 8, key) are trials that
    stop with probability 1/8; a word whose 21 groups all fail is re-mixed (at most 8 words)."""
    h = rnd(seed, 8, np.asarray(keys, dtype=np.uint64))
    cnt = np.ones(len(h), dtype=np.uint64)
    done = np.zeros(len(h), dtype=bool)
    for _ in range(8):
        for g in range(21):
            stop = ((h >> np.uint64(3 * g)) & np.uint64(7)) == 0
            newly = stop & ~done
            done |= newly
            cnt += (~done).astype(np.uint64)
        h = mix64(h)
    return cnt


# config 3: two sets of 100 M 50-bit keys, half of them shared = two windows of one key sequence
CONFIG3 = dict(n=100_000_000, key_bits=50, seed=1, first_a=0, first_b=50_000_000)
# config 4: 64 sets of 50 M keys drawn without replacement from a shared pool of 200 M, geometric counts (mean 8);
# set s walks the pool with stride CONFIG4["mul"][s % 8] from offset 3 125 000 * s
CONFIG4 = dict(sets=64, n=50_000_000, pool=200_000_000, key_bits=50, seed=100,
               mul=[1, 3, 7, 11, 13, 17, 19, 23])


def config4_set_args(s, scale=1.0):
    """zk_synth_keys / set_keys arguments of set s of config 4 (scale < 1 shrinks pool and sets together)."""
    n, pool = int(CONFIG4["n"] * scale), int(CONFIG4["pool"] * scale)
    pool |= 1                                             # odd pool size: every stride of the table is coprime to it
    while any(pool % m == 0 for m in CONFIG4["mul"] if m > 1):
        pool += 2
    return dict(seed=CONFIG4["seed"], first=0, count=n, key_bits=CONFIG4["key_bits"], mul=CONFIG4["mul"][s % 8],
                add=(pool // 64) * s, mod=pool)


# The named configurations of BASELINE.json / SURVEY.md section 8(d).
CONFIGS = {
    # name: reads, L, K, genome, sub rate, N rate
    "config1": dict(reads=10_000, L=150, K=25, genome=100_000, sub=0.005, n=0.0005),
    "config2": dict(reads=50_000_000, L=150, K=25, genome=100_000_000, sub=0.005, n=0.0005),
    "config5": dict(reads=300_000_000, L=150, K=31, genome=3_100_000_000, sub=0.005, n=0.0005),
}
