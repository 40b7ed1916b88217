"""java.util.Random restated (48-bit LCG), so the bench token stream equals the reference's.

LlamaBench draws its synthetic prompt/generation ids with ``new Random(42).nextInt(vocab)``
(J/bench/LlamaBench.java:188-193); this reproduces that stream exactly.
"""
from __future__ import annotations

_MULT = 0x5DEECE66D
_MASK = (1 << 48) - 1


def _s32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


class JavaRandom:
    def __init__(self, seed: int):
        self.seed = (seed ^ _MULT) & _MASK

    def next(self, bits: int) -> int:
        self.seed = (self.seed * _MULT + 0xB) & _MASK
        return _s32(self.seed >> (48 - bits))

    def next_int(self, bound: int | None = None) -> int:
        if bound is None:
            return self.next(32)
        if bound <= 0:
            raise ValueError("bound must be positive")
        r = self.next(31)
        m = bound - 1
        if bound & m == 0:                       # power of two
            return _s32((bound * r) >> 31)
        u = r
        while True:
            r = u % bound
            if _s32(u - r + m) >= 0:             # rejection on int overflow
                return r
            u = self.next(31)


def bench_tokens(vocab: int, n: int, seed: int = 42):
    rng = JavaRandom(seed)
    return [rng.next_int(vocab) for _ in range(n)]
