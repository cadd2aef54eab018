"""Helpers of the bag-of-token builder (/root/reference/src/ir/retriever/index_utils.py:11-21)."""


def get_first_unique_n(iterable, n):
    """First n distinct elements of `iterable`, in order of first appearance (fewer if it runs out)."""
    kept = {}
    for item in iterable:
        if item not in kept:
            kept[item] = None
            if len(kept) == n:
                break
    yield from kept
