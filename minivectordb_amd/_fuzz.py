"""partial_ratio — host-side string similarity used only by hybrid_rerank_results.

The reference calls ``thefuzz.fuzz.partial_ratio`` (minivectordb/vector_database.py:410-411), an un-vendored
dependency absent from this image; thefuzz delegates to rapidfuzz and rounds to an int.  This restates rapidfuzz's
published algorithm (its pure-Python fallback, ``rapidfuzz/fuzz_py.py: _partial_ratio_impl``): the shorter string is
compared with every window of its own length in the longer one plus the shorter windows hanging over both ends
(windows whose outer character does not occur in the shorter string are skipped, as there), each by the Indel ratio
``2 * LCS / (len(a) + len(b))``; equal-length inputs are tried both ways round.  PARITY UNPINNED: rapidfuzz cannot be
run here; the expected values in tests/test_threads_and_rerank.py are hand-computed from that definition.
O(k) string work after the search; not part of the GPU hot path.
"""


def _lcs_len(pattern_masks, m, text):
    """Length of the longest common subsequence of a pattern (given as per-character bit masks over its m positions)
    and `text`: Hyyro's bit-parallel recurrence on Python integers, O(len(text)) big-int operations."""
    full = (1 << m) - 1
    s = full
    for ch in text:
        match = pattern_masks.get(ch, 0)
        u = s & match
        s = ((s + u) | (s - u)) & full
    return m - bin(s).count("1")


def _impl(shorter, longer):
    """Best Indel ratio (0..1) of `shorter` against the candidate windows of `longer` (len(shorter) <= len(longer))."""
    m, n = len(shorter), len(longer)
    chars = set(shorter)
    masks = {}
    for i, ch in enumerate(shorter):
        masks[ch] = masks.get(ch, 0) | (1 << i)
    best = 0.0

    def offer(sub):
        nonlocal best
        if sub:
            r = 2.0 * _lcs_len(masks, m, sub) / (m + len(sub))
            if r > best:
                best = r

    for i in range(1, m):                 # windows hanging over the left end: longer[:i]
        if longer[i - 1] in chars:
            offer(longer[:i])
            if best == 1.0:
                return best
    for i in range(n - m):                # full-length windows
        if longer[i + m - 1] in chars:
            offer(longer[i:i + m])
            if best == 1.0:
                return best
    for i in range(n - m, n):             # the last full window and the ones hanging over the right end
        if longer[i] in chars:
            offer(longer[i:])
            if best == 1.0:
                return best
    return best


def partial_ratio(s1, s2):
    if s1 is None or s2 is None:
        return 0
    if len(s1) == 0 or len(s2) == 0:
        return 100 if len(s1) == len(s2) else 0
    shorter, longer = (s1, s2) if len(s1) <= len(s2) else (s2, s1)
    best = _impl(shorter, longer)
    if best != 1.0 and len(s1) == len(s2):
        best = max(best, _impl(longer, shorter))
    return int(round(100.0 * best))
