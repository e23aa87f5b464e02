"""CPU model of k_pyramid_one's tile algorithm (textslam_amd/csrc/tsorb.hip) against the oracle's pyramid (ORBextractor.cc:1118-1143).

The kernel forms a tile of level l from the INPUT image: the image region under the tile, then levels 1 .. l-1 of exactly the pixels the next level
reads, then the tile.  What must hold for that to give the chained cv::resize's bytes: (1) the needed-range walk (the bordered tile reflects to an interior
range of level l; an interior range [a, b] of level k reads the columns sx(a) .. sx1(b) of level k-1) covers every source pixel of every stage, and (2) the
regions stay inside the buffers the host sized the tiles for.  This file restates the walk in numpy with the kernel's table arithmetic and checks both on
whole levels, tile by tile, including the tiles on the reflected frame.  (The HIP kernel itself is checked bit for bit in tests/test_gpu_orb.py.)"""
import math
import numpy as np
import pytest

from textslam_amd.orbextractor import synthetic_frame

EDGE, P1_BUF, P1_ENT = 19, 16384, 512


def _geometry(w, h, nlevels=8, scale=1.2):
    sf = [np.float32(1.0)]
    for _ in range(1, nlevels):
        sf.append(np.float32(sf[-1] * np.float32(scale)))
    isf = [np.float32(1.0) / s for s in sf]
    rnd = lambda v: int(np.rint(np.float32(v)))
    return [(rnd(np.float32(w) * isf[l]), rnd(np.float32(h) * isf[l])) for l in range(nlevels)]


def _reflect(x, n):
    x = np.where(x < 0, -x, x)
    return np.where(x >= n, 2 * (n - 1) - x, x)


def _xent(dx, scale, Sw):
    fx = (((dx.astype(np.float64) + 0.5) * scale) - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64); fx = fx - sx.astype(np.float32)
    lo = sx < 0; fx = np.where(lo, np.float32(0), fx); sx = np.where(lo, 0, sx)
    hi = sx >= Sw - 1; fx = np.where(hi, np.float32(0), fx); sx = np.where(hi, Sw - 1, sx)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64); a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    return sx, np.where(sx + 1 < Sw, sx + 1, sx), a0, a1


def _yent(dy, scale, Sh):
    fy = (((dy.astype(np.float64) + 0.5) * scale) - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64); fy = fy - sy.astype(np.float32)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64); b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    return np.clip(sy, 0, Sh - 1), np.clip(sy + 1, 0, Sh - 1), b0, b1


def _reflect_range(lo, hi, n):
    a, b = int(_reflect(np.int64(lo), n)), int(_reflect(np.int64(hi), n))
    rlo, rhi = min(a, b), max(a, b)
    if lo <= 0 <= hi: rlo = 0
    if lo <= n - 1 <= hi: rhi = n - 1
    return rlo, rhi


def _stage(src, ox, oy, xe, ye):
    """src: region of the level before with origin (ox, oy); xe / ye: entries of the destination columns / rows"""
    sx, sx1, a0, a1 = xe; sy0, sy1, b0, b1 = ye
    assert sx.min() - ox >= 0 and sx1.max() - ox < src.shape[1] and sy0.min() - oy >= 0 and sy1.max() - oy < src.shape[0]     # (1): covered
    r0, r1 = src[sy0 - oy].astype(np.int64), src[sy1 - oy].astype(np.int64)
    S0 = r0[:, sx - ox] * a0 + r0[:, sx1 - ox] * a1; S1 = r1[:, sx - ox] * a0 + r1[:, sx1 - ox] * a1
    return (((((b0[:, None] * (S0 >> 4)) >> 16) + ((b1[:, None] * (S1 >> 4)) >> 16) + 2) >> 2) & 255).astype(np.uint8)


def tile_from_image(img, geo, l, X0, Y0, TW, TH, stats):
    rs = [None] + [(1.0 / (geo[k][0] / geo[k - 1][0]), 1.0 / (geo[k][1] / geo[k - 1][1])) for k in range(1, len(geo))]
    w, h = geo[l]
    rng = {}
    xr = _reflect_range(X0 - EDGE, X0 + TW - 1 - EDGE, w); yr = _reflect_range(Y0 - EDGE, Y0 + TH - 1 - EDGE, h)
    vx, vy = list(xr), list(yr)
    for k in range(l, 0, -1):                                   # the four range ends, level by level
        ex = _xent(np.array(vx), rs[k][0], geo[k - 1][0]); ey = _yent(np.array(vy), rs[k][1], geo[k - 1][1])
        vx = [int(ex[0][0]), int(ex[1][1])]; vy = [int(ey[0][0]), int(ey[1][1])]
        rng[k - 1] = (vx[0], vx[1], vy[0], vy[1])
    ax, bx, ay, by = rng[0]
    reg = img[ay:by + 1, ax:bx + 1]; ox, oy = ax, ay
    stats["buf"] = max(stats["buf"], 4 * ((bx - ax + 4) // 4) * (by - ay + 1)); nx = ny = 0
    for k in range(1, l):
        ax, bx, ay, by = rng[k]
        reg = _stage(reg, ox, oy, _xent(np.arange(ax, bx + 1), rs[k][0], geo[k - 1][0]), _yent(np.arange(ay, by + 1), rs[k][1], geo[k - 1][1]))
        ox, oy = ax, ay; nx += bx - ax + 1; ny += by - ay + 1
        stats["buf"] = max(stats["buf"], 4 * ((bx - ax + 4) // 4) * (by - ay + 1))
    stats["ent"] = max(stats["ent"], nx + TW, ny + TH)
    return _stage(reg, ox, oy, _xent(_reflect(np.arange(X0, X0 + TW) - EDGE, w), rs[l][0], geo[l - 1][0]),
                  _yent(_reflect(np.arange(Y0, Y0 + TH) - EDGE, h), rs[l][1], geo[l - 1][1]))


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (480, 752)])
def test_tiles_formed_from_the_input_image_equal_the_chained_pyramid(oracle_lib, shape):
    h0, w0 = shape
    img = np.ascontiguousarray(np.tile(synthetic_frame(77), (1, 2))[:h0, :w0])
    geo = _geometry(w0, h0)
    stats = dict(buf=0, ent=0)
    for l in range(1, 8):
        ref = oracle_lib.orb_level(img, l)
        w, h = geo[l]; assert ref.shape == (h + 38, w + 38)
        tw, th = (64 if l <= 2 else 32 if l == 3 else 16), 16
        tiles = [(X0, Y0) for Y0 in range(0, h + 38, th) for X0 in range(0, w + 38, tw)]
        if l < 6:                                               # the shallow levels: the frame's tiles, the corners and a sample of the interior (the model is slow)
            ncol = (w + 38 + tw - 1) // tw; nrow = (h + 38 + th - 1) // th
            tiles = [t for i, t in enumerate(tiles) if t[0] // tw in (0, 1, ncol - 2, ncol - 1) or t[1] // th in (0, 1, nrow - 2, nrow - 1) or i % 17 == 0]
        for X0, Y0 in tiles:
            TW, TH = min(tw, w + 38 - X0), min(th, h + 38 - Y0)
            got = tile_from_image(img, geo, l, X0, Y0, TW, TH, stats)
            assert np.array_equal(got, ref[Y0:Y0 + TH, X0:X0 + TW]), (l, X0, Y0)
    assert stats["buf"] <= P1_BUF and stats["ent"] <= P1_ENT, stats             # (2): what the host's upper bound promises the kernel
