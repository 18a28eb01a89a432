"""
Host-side closed forms behind `zot jaccard -p`: the regularised incomplete beta function by its
series, and its quantiles by bisection.  Plain double arithmetic, evaluated in the same order as the
reference so the printed digits agree (zotmer/library/stats.py:36-129, zotmer/commands/jaccard.py:56-88).

This file is a RESTATEMENT, by necessity close to its source: `log_ix` and `quant_beta` follow `logIx` and `quantBeta`
(jaccard.py:56-84) term by term, because `zot jaccard` prints the results with `%f` and the last digits depend on the
order of the floating-point operations (the series' term recurrence, the bisection's midpoint and stopping rule).  It is
host-side formatting logic, not part of the accelerated path; parity is pinned by tests/golden/f3_jaccard.json.
"""
import math

_LOG_SMALL_FAC = [math.log(math.factorial(n)) for n in range(25)]      # stats.py:45


def log_fac(n):
    """log(n!) -- table below 25, a Ramanujan-type closed form above (stats.py:78-84)."""
    if n < len(_LOG_SMALL_FAC):
        return _LOG_SMALL_FAC[n]
    return n * math.log(n) - n + math.log(n * (1 + 4 * n * (1 + 2 * n))) / 6.0 + math.log(math.pi) / 2.0


def log_add(a, b):
    """log(exp(a) + exp(b))  (stats.py:86-93)"""
    hi, lo = max(a, b), min(a, b)
    return hi + math.log1p(math.exp(lo - hi))


def log_choose(n, k):
    """log C(n, k)  (stats.py:121-128)"""
    if k == 0 or k == n:
        return 0
    return log_fac(n) - (log_fac(n - k) + log_fac(k))


def log_ix(x, m, n):
    """log I_x(m, n) through sum_{j>=m} C(n+j-1, j) x^j (1-x)^n, summed until it stops moving
    (jaccard.py:56-71)."""
    lx = math.log(x)
    j = m
    v = log_choose(n + j - 1, j)
    s = v + j * lx
    while True:
        j += 1
        v += math.log((n + j - 1.0) / j)
        u = log_add(s, v + j * lx)
        if u == s:
            break
        s = u
    return n * math.log1p(-x) + s


def quant_beta(q, m, n):
    """The q-quantile of Beta(m, n) by bisection to 1e-7; the LOWER end is returned (jaccard.py:73-84)."""
    lq = math.log(q)
    lo, hi = 1e-10, 1 - 1e-10
    while (hi - lo) > 1e-7:
        x = (hi + lo) / 2.0
        if log_ix(x, m, n) < lq:
            lo = x
        else:
            hi = x
    return lo


def jaccard_fields(nx, ny, isec, p=None):
    """The numeric columns of one `zot jaccard` line (jaccard.py:128-134, 155-161) as one string."""
    union = nx + ny - isec
    d = float(isec) / float(union)
    if p is None:
        return "%d\t%d\t%d\t%d\t%f" % (nx, ny, isec, union, d)
    m, n = isec + 1, (union - isec) + 1
    pv = log_ix(p, m, n) / math.log(10)
    q05 = quant_beta(0.05, m, n)
    q95 = quant_beta(0.95, m, n)
    return "%d\t%d\t%d\t%d\t%f\t-%f\t+%f\t%f" % (nx, ny, isec, union, d, d - q05, q95 - d, pv)
