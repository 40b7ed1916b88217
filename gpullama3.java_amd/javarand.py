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


# ---------------------------------------------------------------------------------------------------------------------
# RandomGeneratorFactory.getDefault().create(seed) = L32X64MixRandom (the JDK 17+ default algorithm), which
# Sampler.selectSampler (J/inference/sampler/Sampler.java:84) hands to CategoricalSampler / ToppSampler.  The HIP library
# never draws random numbers itself: the host passes rng.nextFloat(1f) into gl3_forward_decode_sample, so a Java host uses the
# JDK's own generator.  This Python twin exists for the Python host mirror and the tests only.
# UNPINNED: restated from the published algorithm (JDK java.base jdk.internal.random.L32X64MixRandom / RandomSupport:
# LCG multiplier 0xadb4a92d, xoroshiro64 (26, 9, 13), mixLea32 output function, seed expansion with mixMurmur32 / mixLea32 over
# SILVER_RATIO_64 / GOLDEN_RATIO_32) without a JDK at hand to produce known answers; if it disagreed with a real JVM the
# sampled ids of the Python mirror would differ from the Java host's, never the library's arithmetic.
_M32 = 0xFFFFFFFF


def _mix_murmur32(z: int) -> int:
    z &= _M32
    z = ((z ^ (z >> 16)) * 0x85EBCA6B) & _M32
    z = ((z ^ (z >> 13)) * 0xC2B2AE35) & _M32
    return z ^ (z >> 16)


def _mix_lea32(z: int) -> int:
    z &= _M32
    z = ((z ^ (z >> 16)) * 0xD36D884B) & _M32
    z = ((z ^ (z >> 16)) * 0xD36D884B) & _M32
    return z ^ (z >> 16)


def _rotl32(x: int, k: int) -> int:
    return ((x << k) | (x >> (32 - k))) & _M32


class L32X64MixRandom:
    _M = 0xADB4A92D
    _GOLDEN_32 = 0x9E3779B9
    _SILVER_64 = 0x6A09E667F3BCC909

    def __init__(self, seed: int):
        seed = (seed ^ self._SILVER_64) & 0xFFFFFFFFFFFFFFFF
        self.a = _mix_murmur32(seed >> 32) | 1
        self.s = 1
        self.x0 = _mix_lea32(seed & _M32)
        self.x1 = _mix_lea32((seed + self._GOLDEN_32) & _M32)
        if (self.x0 | self.x1) == 0:
            v = (self.s + self._GOLDEN_32) & _M32
            self.x0 = _mix_murmur32(v)
            self.x1 = _mix_murmur32((v + self._GOLDEN_32) & _M32)

    def next_int(self) -> int:
        result = _mix_lea32((self.s + self.x0) & _M32)
        self.s = (self._M * self.s + self.a) & _M32
        q0, q1 = self.x0, self.x1
        q1 ^= q0
        q0 = _rotl32(q0, 26)
        q0 = (q0 ^ q1 ^ (q1 << 9)) & _M32
        q1 = _rotl32(q1, 13)
        self.x0, self.x1 = q0, q1
        return result

    def next_float(self) -> float:
        """RandomGenerator.nextFloat(1f) = RandomSupport.boundedNextFloat: (nextInt() >>> 8) * 2^-24 (always < 1)."""
        return (self.next_int() >> 8) * (1.0 / (1 << 24))
