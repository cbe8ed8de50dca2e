"""Seeded synthetic inputs shared by tests and bench.py (numpy PCG64; the reference's
rand 0.7.3 StdRng streams cannot be regenerated without Rust -- SURVEY.md 8c).

Distributions follow the reference's bench generators (benches/rand_benchmarks.rs):
  * "random": bytes i.i.d. uniform on 1..=255 (no NUL so the data is legal for hamming_search)
  * "mutated": b = a with U[k/2, k] random edits {substitute->0x20, insert, delete} (:207-238),
    alphabet 33..=126 (:241); optional adjacent swaps for the transposition path.
"""
import numpy as np


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def random_bytes(g, n, lo=1, hi=256):
    return g.integers(lo, hi, size=n, dtype=np.uint8)


def rand_str(g, n):
    return g.integers(33, 127, size=n, dtype=np.uint8).tobytes()


def mutate(g, a: bytes, k: int, swaps=False) -> bytes:
    """<= k edits (each of cost <= 1 under LEVENSHTEIN/RDAMERAU costs)."""
    n_edits = int(g.integers(k // 2, k + 1)) if k > 0 else 0
    s = bytearray(a)
    for _ in range(n_edits):
        kinds = 4 if swaps else 3
        t = int(g.integers(0, kinds))
        if t == 0 and len(s) > 0:
            s[int(g.integers(0, len(s)))] = 32
        elif t == 1:
            s.insert(int(g.integers(0, len(s) + 1)), int(g.integers(33, 127)))
        elif t == 2 and len(s) > 0:
            del s[int(g.integers(0, len(s)))]
        elif t == 3 and len(s) > 1:
            p = int(g.integers(0, len(s) - 1))
            s[p], s[p + 1] = s[p + 1], s[p]
    return bytes(s)


def pairs_random(seed, n, length):
    g = rng(seed)
    a = random_bytes(g, n * length).reshape(n, length)
    b = random_bytes(g, n * length).reshape(n, length)
    return a, b


def pairs_mutated_fixed(seed, n, length, k, swaps=False):
    """Fixed-length mutated pairs (b truncated / padded back to `length`) as (n,length) arrays."""
    g = rng(seed)
    a = g.integers(33, 127, size=(n, length), dtype=np.uint8)
    b = np.empty_like(a)
    for i in range(n):
        m = mutate(g, a[i].tobytes(), k, swaps)
        m = (m + rand_str(g, length))[:length]
        b[i] = np.frombuffer(m, dtype=np.uint8)
    return a, b


def planted_haystack(seed, needle: bytes, haystack_len, every, k):
    """Random printable haystack with a mutated needle copy planted about every `every` bytes."""
    g = rng(seed)
    h = bytearray(g.integers(33, 127, size=haystack_len, dtype=np.uint8).tobytes())
    pos = every // 2
    while pos + 2 * len(needle) < haystack_len:
        m = mutate(g, needle, k)
        h[pos:pos + len(m)] = m
        pos += every
    return bytes(h[:haystack_len])
