"""``partition`` as used by the BVH builder (role of src/stl4py.py:26-61)."""


def partition(iterable, pred, first=0, last=None):
    """In-place unstable partition of iterable[first:last]; returns the index one
    past the last element satisfying ``pred``."""
    hi = len(iterable) if last is None else last
    lo = first
    while True:
        while lo < hi and pred(iterable[lo]):
            lo += 1
        while lo < hi and not pred(iterable[hi - 1]):
            hi -= 1
        if hi - lo < 2:
            return lo
        iterable[lo], iterable[hi - 1] = iterable[hi - 1], iterable[lo]
        lo += 1
        hi -= 1
