"""
Distances between k-mer sets from the three counts a = |X & Y|, b = |X \\ Y|, c = |Y \\ X|
(the GPU's zk_split).  Reference: zotmer/library/dist.py:17-239 (set form of each measure, formulas
from arXiv 1604.02412 table 1) and the registry in zotmer/commands/dist.py:59-92.

Each formula keeps the reference's operation order, so the doubles are the same bit for bit.
The "quant" / "ab" / jensen.shannon entries are vector measures over 4**K counters; the reference's
own code path for them cannot run and they are listed but refused here.
"""
import math


def _bray_curtis(a, b, c):      # dist.py:40-41 (sorensen :209-210 is the same expression)
    return float(b + c) / float(2 * a + b + c)


def _chord(a, b, c):            # dist.py:67-68 (hellinger :93-94 is the same expression)
    return math.sqrt(2 * (1 - a / math.sqrt((a + b) * (a + c))))


def _jaccard(a, b, c):          # dist.py:112-113
    return float(b + c) / float(a + b + c)


def _kulczynski(a, b, c):       # dist.py:168-172
    a, b, c = float(a), float(b), float(c)
    return 1 - 0.5 * (a / (a + b) + a / (a + c))


def _ochiai(a, b, c):           # dist.py:190-191
    return 1 - a / math.sqrt((a + b) * (a + c))


def _whittaker(a, b, c):        # dist.py:235-239
    a, b, c = float(a), float(b), float(c)
    return 0.5 * (b / (a + b) + c / (a + c) + abs(a / (a + b) - a / (a + c)))


# name -> (description, is_vector, function of (a, b, c) or None)      commands/dist.py:59-92
MEASURES = {
    "bray.curtis.quant": ("Quantative Bray.Curtis distance", True, None),
    "bray.curtis.qual": ("Qualitative Bray.Curtis distance", False, _bray_curtis),
    "chord.quant": ("Quantative Chord distance", True, None),
    "chord.qual": ("Qualitative Chord distance", False, _chord),
    "hellinger.quant": ("Quantative Hellinger distance", True, None),
    "hellinger.qual": ("Qualitative Hellinger distance", False, _chord),
    "jaccard.ab": ("Abundance.based Jaccard distance", True, None),
    "jaccard.qual": ("Qualitative Jaccard distance", False, _jaccard),
    "jensen.shannon": ("Jensen.Shannon distance", True, None),
    "kulczynski.quant": ("Quantative Kulczynski distance", True, None),
    "kulczynski.qual": ("Qualitative Kulczynski distance", False, _kulczynski),
    "ochiai.ab": ("Abundance.based Ochiai distance", True, None),
    "ochiai.qual": ("Qualitative Ochiai distance", False, _ochiai),
    "sorensen.ab": ("Abundance.based Sorensen distance", True, None),
    "sorensen.qual": ("Qualitative Sorensen distance", False, _bray_curtis),
    "whittaker.quant": ("Quantative Whittaker distance", True, None),
    "whittaker.qual": ("Qualitative Whittaker distance", False, _whittaker),
}
