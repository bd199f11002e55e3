"""64-bit order-sensitive digest of a small integer/float array, cheap in numpy and in C.

    h(w[0..n)) = ( sum_i (w_i + 1) * P^(i+1) mod 2^64 ) xor (n * Q mod 2^64)

over the array viewed as little-endian 32-bit words. Used to pin observations in the golden
fixtures without storing them (tests/golden) and mirrored in oracle/digest.h.
"""
from __future__ import annotations

import numpy as np

P = np.uint64(0x9E3779B97F4A7C15)
Q = np.uint64(0xC2B2AE3D27D4EB4F)


def digest_words(arr: np.ndarray) -> int:
    w = np.ascontiguousarray(arr).view(np.uint32).ravel().astype(np.uint64)
    n = w.size
    if n == 0:
        return 0
    with np.errstate(over="ignore"):
        pw = np.cumprod(np.full(n, P, dtype=np.uint64))
        h = np.sum((w + np.uint64(1)) * pw, dtype=np.uint64)
        h = h ^ (np.uint64(n) * Q)
    return int(h)


def splitmix64(x: int) -> int:
    """the public-domain splitmix64 finaliser; drives the build's counter-based test policies."""
    m = (1 << 64) - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)
