"""Deterministic synthetic inputs: the integer-only generators the golden fixtures, the tests, bench.py
and the tools all draw their images from.  Pure numpy; no codec code and nothing of the checker in here.

  g1       (x*7 + y*13 + k*37 + ((x*y)&3)*5) & 0xFF
  g2       (x + y + 37*k) & 0xFF                      -- clean gradient, the compressible extreme
  g3       std::mt19937(seed), one draw per sample   -- noise, the incompressible extreme
  mid      g2 + a small integer dither in [-8, 7]
  checker  saturated 0 / 255 checkerboard
  nat      photo-like: smooth low-frequency shading, blocky objects with hard edges, +-1 sensor-like noise
"""
import numpy as np


def gen_g1(w, h, c):
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    return ((x * 7 + y * 13 + k * 37 + ((x * y) & 3) * 5) & 0xFF).astype(np.uint8)


def gen_g2(w, h, c):
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    return ((x + y + 37 * k) & 0xFF).astype(np.uint8)


def gen_g3(w, h, c, seed=1234):
    """std::mt19937(seed), one draw per sample in (y,x,k) order, & 0xFF (numpy's MT19937 is the same
    generator; random_raw yields the same 32-bit outputs as std::mt19937::operator())."""
    from numpy.random import MT19937

    bg = MT19937()
    # seed exactly like std::mt19937(seed): the classic init_genrand recurrence
    st = np.zeros(624, dtype=np.uint32)
    st[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        st[i] = (1812433253 * (int(st[i - 1]) ^ (int(st[i - 1]) >> 30)) + i) & 0xFFFFFFFF
    bg.state = {"bit_generator": "MT19937", "state": {"key": st, "pos": 624}}
    raw = bg.random_raw(w * h * c)
    return (raw & 0xFF).astype(np.uint8).reshape(h, w, c)


def gen_mid(w, h, c, seed=7):
    """mid-entropy integer pattern: G2 gradient + small LCG dither in [-8,7] (no floating point)."""
    n = w * h * c
    a = np.arange(n, dtype=np.uint64)
    # closed-form-free LCG via vectorised hash of the index (splitmix-like, integer only)
    z = (a + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    d = ((z >> np.uint64(60)).astype(np.int64) - 8).reshape(h, w, c)
    return ((gen_g2(w, h, c).astype(np.int64) + d) & 0xFF).astype(np.uint8)


def gen_checker(w, h, c):
    y, x, k = np.meshgrid(np.arange(h), np.arange(w), np.arange(c), indexing="ij")
    return (((x + y + k) & 1) * 255).astype(np.uint8)


def gen_nat(w, h, c, seed=11):
    """photo-like integer pattern: slow quadratic shading + 64-pixel blocks with hard edges + noise in [-1, 1]."""
    x = np.arange(w, dtype=np.int32)[None, :, None]
    y = np.arange(h, dtype=np.int32)[:, None, None]
    k = np.arange(c, dtype=np.int32)[None, None, :]
    shade = ((x * x) >> 12) + ((y * 3) >> 2) + 29 * k
    blocks = 45 * (((x >> 6) * 7 + (y >> 6) * 13 + k) % 3)
    z = (np.arange(w * h * c, dtype=np.uint32) + np.uint32(seed)) * np.uint32(0x9E3779B1)  # Knuth's multiplicative hash
    z ^= z >> np.uint32(15)
    z *= np.uint32(0x85EBCA6B)
    noise = (z >> np.uint32(30)).astype(np.int32).reshape(h, w, c) - 1  # -1, 0, 1, 2
    noise[noise == 2] = 0
    return ((shade + blocks + noise) & 0xFF).astype(np.uint8)


GENERATORS = {"g1": gen_g1, "g2": gen_g2, "g3": gen_g3, "mid": gen_mid, "checker": gen_checker, "nat": gen_nat}
