#!/usr/bin/env python3
"""Lock-step length of the slice kernels on a given content (numpy, no GPU): all 64 lanes of a wavefront walk the coding
phases of a sample together, so a wavefront-sample costs 2*max(exponent)+3 bin slots where a lone lane would need
2*exponent+3 (llcomp.hpp:166-206: zero flag, unary exponent, mantissa, sign).  Prints, for planar one-row slices, the sum
over wavefront-samples of the lock-step bins against the mean bins per lane -- for slices dealt to wavefronts in container
order (what the kernels do) and for slices sorted by activity first (a slice -> lane permutation).

    python tools/lockstep_analysis.py [tile_w=480] [contents...]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llcomp_amd import synth  # noqa: E402


def planes(img):
    """colour transform of llcomp.hpp:396-409 -> int planes [c][h][w] in coding order (r-g, g', b-g)"""
    p = img.astype(np.int32)
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    b = b - g
    r = r - g
    s = b + r
    g = g + np.where(s >= 0, s // 4, -((-s) // 4))  # truncating division
    return np.stack([r, g, b])


def exponents(pl, tile_w):
    """[slices][tile_w] exponent of |residual| per sample (-1 for a zero residual) of planar one-row slices, slice order
    = (y, tx, channel) as in the container"""
    c, h, w = pl.shape
    ntx = (w + tile_w - 1) // tile_w
    assert w % tile_w == 0, "this analysis takes whole tiles only"
    t = pl.reshape(c, h, ntx, tile_w)
    left = np.concatenate([np.full((c, h, ntx, 1), 128, np.int32), t[..., :-1]], axis=3)  # slice-local border: l = 128 at x == 0
    res = t - left
    a = np.abs(res)
    e = np.where(a > 0, np.floor(np.log2(np.maximum(a, 1))).astype(np.int32), -1)
    return e.transpose(1, 2, 0, 3).reshape(h * ntx * c, tile_w)


def lockstep(e):
    """(lock-step bins, mean bins per lane) summed over wavefront-samples, 64 slices per wavefront in the given order"""
    n = e.shape[0] // 64 * 64
    g = e[:n].reshape(-1, 64, e.shape[1])
    bins = np.where(g >= 0, 2 * g + 3, 1)
    mx = g.max(axis=1)
    lock = np.where(mx >= 0, 2 * mx + 3, 1)
    return float(lock.sum()), float(bins.mean(axis=1).sum())


def main():
    tile_w = int(sys.argv[1]) if len(sys.argv) > 1 else 480
    contents = sys.argv[2:] or ["g3", "mid", "nat", "g2"]
    w, h, c = 3840, 2160, 3
    print(f"4K RGB8, planar {tile_w}x1 slices, 64 slices per wavefront; bins per wavefront-sample")
    print(f"{'content':8s} {'mean/lane':>10s} {'lock-step':>10s} {'overhead':>9s} | {'sorted by activity':>18s} {'overhead':>9s} {'gain':>6s}")
    for name in contents:
        img = synth.GENERATORS[name](w, h, c)
        e = exponents(planes(img), tile_w)
        lock, mean = lockstep(e)
        act = np.where(e >= 0, 2 * e + 3, 1).sum(axis=1)
        order = np.argsort(act, kind="stable")
        lock_s, mean_s = lockstep(e[order])
        nws = e.shape[0] // 64 * e.shape[1]
        print(f"{name:8s} {mean / nws:10.2f} {lock / nws:10.2f} {lock / mean - 1:9.1%} | {lock_s / nws:18.2f} {lock_s / mean_s - 1:9.1%} {1 - lock_s / lock:6.1%}")


if __name__ == "__main__":
    main()


def regroup_report(tile_w=480):
    """slices of one (tile column, channel) over 64 consecutive rows in one wavefront: lanes then look at the same image
    column at every step"""
    w, h, c = 3840, 2160, 3
    ntx = w // tile_w
    print(f"\nwavefront = 64 vertically adjacent rows of one (tile column, channel) instead of container order ({tile_w}x1):")
    for name in ["g3", "mid", "nat", "g2"]:
        img = synth.GENERATORS[name](w, h, c)
        e = exponents(planes(img), tile_w)          # [(y, tx, ch)][k]
        lock, mean = lockstep(e)
        ev = e.reshape(h, ntx, c, tile_w)
        hh = h // 64 * 64
        v = ev[:hh].transpose(1, 2, 0, 3).reshape(ntx * c * hh, tile_w)   # (tx, ch, y)
        lock_v, mean_v = lockstep(v)
        lock0, mean0 = lockstep(ev[:hh].reshape(hh * ntx * c, tile_w))
        print(f"  {name:5s} container order {lock0 / mean0:5.2f}x mean  ->  vertical groups {lock_v / mean_v:5.2f}x mean   ({1 - lock_v / lock0:+.1%} lock-step bins)")


if __name__ == "__main__" and len(sys.argv) <= 1:
    regroup_report()


def channel_split_report(tile_w=480):
    """one channel per wavefront: 64 consecutive tiles (row-major) of the same channel"""
    w, h, c = 3840, 2160, 3
    ntx = w // tile_w
    print(f"\nwavefront = 64 consecutive tiles of ONE channel instead of 64 consecutive (tile, channel) slices ({tile_w}x1):")
    for name in ["g3", "mid", "nat", "g2"]:
        img = synth.GENERATORS[name](w, h, c)
        e = exponents(planes(img), tile_w)          # [(y, tx, ch)][k]
        lock0, mean0 = lockstep(e)
        v = e.reshape(h * ntx, c, tile_w).transpose(1, 0, 2).reshape(c * h * ntx, tile_w)
        lock_v, mean_v = lockstep(v)
        per_ch = [lockstep(v[i * h * ntx:(i + 1) * h * ntx]) for i in range(c)]
        print(f"  {name:5s} container order {lock0 / mean0:5.2f}x mean  ->  one channel per wavefront {lock_v / mean_v:5.2f}x mean   ({lock_v / lock0 - 1:+.1%} lock-step bins; "
              + ", ".join(f"ch{i}: {a / (h * ntx // 64 * tile_w):.2f}" for i, (a, b) in enumerate(per_ch)) + ")")


if __name__ == "__main__" and len(sys.argv) <= 1:
    channel_split_report()
