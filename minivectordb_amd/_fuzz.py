"""partial_ratio — host-side string similarity used only by hybrid_rerank_results.

The reference calls ``thefuzz.fuzz.partial_ratio`` (minivectordb/vector_database.py:410-411), an
un-vendored dependency absent from this image.  This is the classic difflib formulation of that
score (best ratio of the shorter string against equally long windows of the longer one, aligned
on matching blocks), 0..100.  O(k) string work after the search; not part of the GPU hot path.
"""
from difflib import SequenceMatcher


def partial_ratio(s1, s2):
    if s1 is None or s2 is None or len(s1) == 0 or len(s2) == 0:
        return 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    blocks = SequenceMatcher(None, shorter, longer, autojunk=False).get_matching_blocks()
    best = 0.0
    for a, b, _size in blocks:
        start = max(0, b - a)
        window = longer[start:start + len(shorter)]
        r = SequenceMatcher(None, shorter, window, autojunk=False).ratio()
        if r > 0.995:
            return 100
        best = max(best, r)
    return int(round(100 * best))
