"""The more_itertools functions the reference's tests use (the package is not in this image)."""
from itertools import islice, tee


def take(n, iterable):
    return list(islice(iterable, n))


def partition(pred, iterable):
    a, b = tee(iterable)
    return (x for x in a if not pred(x)), (x for x in b if pred(x))


def consume(iterator, n=None):
    if n is None:
        for _ in iterator:
            pass
    else:
        next(islice(iterator, n, n), None)


def ilen(iterable):
    return sum(1 for _ in iterable)
