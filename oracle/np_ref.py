"""Second, independent restatement (numpy) of the integer/float pixel arithmetic, used ONLY to cross-check the C
oracle for transcription errors ("cross-checked between two independent restatements").
Test infrastructure; deliberately simple and slow -- small inputs only.
"""
from __future__ import annotations

import numpy as np

BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4


def border(p: np.ndarray, n: int, mode: int) -> np.ndarray:
    """cv::borderInterpolate, vectorised; BORDER_CONSTANT -> -1."""
    p = np.asarray(p, np.int64)
    if mode == BORDER_REPLICATE:
        return np.clip(p, 0, n - 1)
    if mode == BORDER_REFLECT:
        if n == 1:
            return np.zeros_like(p)
        m = np.mod(p, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m)
    if mode == BORDER_REFLECT_101:
        if n == 1:
            return np.zeros_like(p)
        m = np.mod(p, 2 * n - 2)
        return np.where(m < n, m, 2 * n - 2 - m)
    if mode == BORDER_WRAP:
        return np.mod(p, n)
    return np.where((p >= 0) & (p < n), p, -1)


def cv_round(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32)
    r = np.rint(x)  # half to even
    bad = ~((r >= -2147483648.0) & (r < 2147483648.0))
    out = np.where(bad, 0, r).astype(np.int64)
    return np.where(bad, -2147483648, out)


def remap_u8(src: np.ndarray, xmap: np.ndarray, ymap: np.ndarray, interp: int, bmode: int) -> np.ndarray:
    """cv::remap for 8-bit images with two float maps (INTER_NEAREST=0 / INTER_LINEAR=1)."""
    h, w = src.shape[:2]
    s = src.reshape(h, w, -1).astype(np.int64)
    if interp == 0:
        sx = np.clip(cv_round(xmap), -32768, 32767)
        sy = np.clip(cv_round(ymap), -32768, 32767)
        inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        if bmode == BORDER_CONSTANT:
            out = np.where(inside[..., None], s[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)], 0)
        else:
            out = s[border(sy, h, bmode), border(sx, w, bmode)]
        return out.astype(np.uint8).reshape(xmap.shape + src.shape[2:])
    isx = cv_round(xmap.astype(np.float32) * np.float32(32))
    isy = cv_round(ymap.astype(np.float32) * np.float32(32))
    ix = np.clip(isx >> 5, -32768, 32767)
    iy = np.clip(isy >> 5, -32768, 32767)
    ax, ay = isx & 31, isy & 31
    acc = np.zeros(xmap.shape + (s.shape[2],), np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            wgt = (ax if dx else 32 - ax) * (ay if dy else 32 - ay) * 32
            xx, yy = ix + dx, iy + dy
            if bmode == BORDER_CONSTANT:
                ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
                v = np.where(ok[..., None], s[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)
            else:
                v = s[border(yy, h, bmode), border(xx, w, bmode)]
            acc += v * wgt[..., None]
    out = (acc + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8).reshape(xmap.shape + src.shape[2:])


def pyr_down(src: np.ndarray) -> np.ndarray:
    """pyrDown for int16 (integer rounding (v+128)>>8) or float32 (scalar association, *1/256)."""
    flt = src.dtype == np.float32
    h, w = src.shape[:2]
    s = src.reshape(h, w, -1)
    s = s.astype(np.float32) if flt else s.astype(np.int64)
    dh, dw = (h + 1) // 2, (w + 1) // 2
    xs = [border(2 * np.arange(dw) - 2 + k, w, BORDER_REFLECT_101) for k in range(5)]
    ys = [border(2 * np.arange(dh) - 2 + k, h, BORDER_REFLECT_101) for k in range(5)]

    def five(t):
        a = t[2] * 6 + (t[1] + t[3]) * 4
        a = a + t[0]
        return a + t[4]

    rows = five([s[:, xs[k]] for k in range(5)])            # h x dw x c
    out = five([rows[ys[k]] for k in range(5)])             # dh x dw x c
    if flt:
        out = (out * np.float32(1.0 / 256)).astype(np.float32)
    else:
        out = ((out + 128) >> 8).astype(np.int16)
    return out.reshape((dh, dw) + src.shape[2:])


def pyr_up(src: np.ndarray) -> np.ndarray:
    """pyrUp to exactly twice the size (int16: (v+32)>>6)."""
    assert src.dtype == np.int16
    h, w = src.shape[:2]
    s = src.reshape(h, w, -1).astype(np.int64)

    def idx(n):
        i = np.arange(n)
        m = i - 1
        m[0] = min(1, n - 1)
        p = i + 1
        p[-1] = n - 1
        return m, i, p

    xm, xc, xp = idx(w)
    he = s[:, xm] + 6 * s[:, xc] + s[:, xp]
    ho = 4 * (s[:, xc] + s[:, xp])
    rows = np.empty((h, 2 * w, s.shape[2]), np.int64)
    rows[:, 0::2], rows[:, 1::2] = he, ho
    ym, yc, yp = idx(h)
    ve = rows[ym] + 6 * rows[yc] + rows[yp]
    vo = 4 * (rows[yc] + rows[yp])
    out = np.empty((2 * h, 2 * w, s.shape[2]), np.int64)
    out[0::2], out[1::2] = ve, vo
    return ((out + 32) >> 6).astype(np.int16).reshape((2 * h, 2 * w) + src.shape[2:])


def trunc_s16(f: np.ndarray) -> np.ndarray:
    return np.trunc(np.asarray(f, np.float32)).astype(np.int64).astype(np.int16)


def multiband_blend(images, masks, tls, roi, num_bands):
    """MultiBandBlender prepare/feed/blend with int16 pyramids and f32 weights (images int16 HxWx3)."""
    x0, y0, w, h = roi
    nb = min(num_bands, int(np.ceil(np.log(max(w, h)) / np.log(2.0))))
    m = 1 << nb
    pw, ph = w + (m - w % m) % m, h + (m - h % m) % m
    lw, lh = [pw], [ph]
    for _ in range(nb):
        lw.append((lw[-1] + 1) // 2)
        lh.append((lh[-1] + 1) // 2)
    lap = [np.zeros((lh[i], lw[i], 3), np.int16) for i in range(nb + 1)]
    wgt = [np.zeros((lh[i], lw[i]), np.float32) for i in range(nb + 1)]
    gap = 3 * m
    for img, mask, (tx, ty) in zip(images, masks, tls):
        ih, iw = mask.shape
        tnx, tny = max(x0, tx - gap), max(y0, ty - gap)
        bnx, bny = min(x0 + pw, tx + iw + gap), min(y0 + ph, ty + ih + gap)
        tnx = x0 + (((tnx - x0) >> nb) << nb)
        tny = y0 + (((tny - y0) >> nb) << nb)
        width, height = bnx - tnx, bny - tny
        width += (m - width % m) % m
        height += (m - height % m) % m
        bnx, bny = tnx + width, tny + height
        dx, dy = max(bnx - (x0 + pw), 0), max(bny - (y0 + ph), 0)
        tnx, bnx, tny, bny = tnx - dx, bnx - dx, tny - dy, bny - dy
        top, left = ty - tny, tx - tnx
        yy = border(np.arange(height) - top, ih, BORDER_REFLECT)
        xx = border(np.arange(width) - left, iw, BORDER_REFLECT)
        g = [img[yy][:, xx].astype(np.int16)]
        wm = np.zeros((height, width), np.float32)
        wm[top:top + ih, left:left + iw] = mask.astype(np.float32) * np.float32(1.0 / 255.0)
        ws = [wm]
        for _ in range(nb):
            g.append(pyr_down(g[-1]))
            ws.append(pyr_down(ws[-1]))
        for i in range(nb):
            up = pyr_up(g[i + 1])
            g[i] = np.clip(g[i].astype(np.int32) - up.astype(np.int32), -32768, 32767).astype(np.int16)
        xtl, ytl, xbr, ybr = tnx - x0, tny - y0, bnx - x0, bny - y0
        for i in range(nb + 1):
            add = trunc_s16(g[i].astype(np.float32) * ws[i][..., None])
            sl = (slice(ytl, ybr), slice(xtl, xbr))
            lap[i][sl] = (lap[i][sl].astype(np.int32) + add.astype(np.int32)).astype(np.int16)
            wgt[i][sl] = wgt[i][sl] + ws[i]
            xtl, ytl, xbr, ybr = xtl // 2, ytl // 2, xbr // 2, ybr // 2
    for i in range(nb + 1):
        lap[i] = trunc_s16(lap[i].astype(np.float32) / (wgt[i] + np.float32(1e-5))[..., None])
    for i in range(nb, 0, -1):
        up = pyr_up(lap[i])
        lap[i - 1] = np.clip(up.astype(np.int32) + lap[i - 1].astype(np.int32), -32768, 32767).astype(np.int16)
    mask_out = np.where(wgt[0][:h, :w] > np.float32(1e-5), 255, 0).astype(np.uint8)
    res = lap[0][:h, :w].copy()
    res[mask_out == 0] = 0
    return res, mask_out


def distance_l1(mask: np.ndarray) -> np.ndarray:
    """Exact city-block distance to the nearest zero pixel; 65534 when the image holds none."""
    h, w = mask.shape
    zy, zx = np.nonzero(mask == 0)
    if len(zy) == 0:
        return np.full((h, w), 65534.0, np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    d = np.abs(yy[..., None] - zy[None, None, :]) + np.abs(xx[..., None] - zx[None, None, :])
    return d.min(axis=2).astype(np.float32)


def dilate3(mask: np.ndarray) -> np.ndarray:
    h, w = mask.shape
    p = np.zeros((h + 2, w + 2), mask.dtype)
    p[1:-1, 1:-1] = mask
    out = np.zeros_like(mask)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, p[dy:dy + h, dx:dx + w])
    return out


def resize_linear_exact(src: np.ndarray, dsize) -> np.ndarray:
    sh, sw = src.shape
    dw, dh = dsize

    def coeffs(ssize, dsz):
        scale = 1.0 / (dsz / ssize)
        ofs = np.zeros(dsz, np.int64)
        c1 = np.zeros(dsz, np.int64)
        for d in range(dsz):
            f = scale * (d + 0.5) - 0.5
            i = int(np.floor(f))
            if i >= 0 and ssize > 1:
                if i < ssize - 1:
                    ofs[d], c1[d] = i, int(np.rint((f - i) * 256.0))
                else:
                    ofs[d], c1[d] = ssize - 1, -1
            else:
                ofs[d], c1[d] = 0, -1
        return ofs, c1

    xo, xc = coeffs(sw, dw)
    yo, yc = coeffs(sh, dh)
    s = src.astype(np.int64)
    out = np.zeros((dh, dw), np.uint8)
    for y in range(dh):
        r0 = s[yo[y]]
        r1 = s[yo[y] + 1] if yc[y] >= 0 else r0
        cy1 = max(yc[y], 0)
        for x in range(dw):
            if xc[x] >= 0:
                h0 = r0[xo[x]] * (256 - xc[x]) + r0[xo[x] + 1] * xc[x]
                h1 = r1[xo[x]] * (256 - xc[x]) + r1[xo[x] + 1] * xc[x]
            else:
                h0, h1 = r0[xo[x]] << 8, r1[xo[x]] << 8
            out[y, x] = (h0 * (256 - cy1) + h1 * cy1 + (1 << 15)) >> 16
    return out


def _area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab (imgproc/resize.cpp): per destination index the covered source cells and their share."""
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            tab.append((d, s1 - 1, np.float32((s1 - f1) / cell)))
        for sx in range(s1, s2):
            tab.append((d, sx, np.float32(1.0 / cell)))
        if f2 - s2 > 1e-3:
            tab.append((d, s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
    return tab


def resize_area(src: np.ndarray, fx: float, fy: float) -> np.ndarray:
    """cv.resize(src, None, fx=fx, fy=fy, interpolation=INTER_AREA), decimation; float32 accumulation in table order."""
    h, w = src.shape[:2]
    s3 = src.reshape(h, w, -1)
    cn = s3.shape[2]
    dw, dh = int(np.rint(w * fx)), int(np.rint(h * fy))
    sx, sy = 1.0 / fx, 1.0 / fy
    isx, isy = int(np.rint(sx)), int(np.rint(sy))
    eps = np.finfo(np.float64).eps
    out = np.zeros((dh, dw, cn), np.uint8)
    if abs(sx - isx) < eps and abs(sy - isy) < eps:
        for dy in range(dh):
            for dx in range(dw):
                blk = s3[dy * isy:(dy + 1) * isy, dx * isx:(dx + 1) * isx].astype(np.int64)
                cnt = blk.shape[0] * blk.shape[1]
                if cnt == 0:
                    continue
                tot = blk.sum(axis=(0, 1))
                full = blk.shape[0] == isy and dx < w // isx
                if full and isx == 2 and isy == 2:
                    out[dy, dx] = (tot + 2) >> 2
                elif full:
                    out[dy, dx] = np.clip(cv_round(tot.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(isx * isy))), 0, 255)
                else:
                    out[dy, dx] = np.clip(cv_round(tot.astype(np.float32) / np.float32(cnt)), 0, 255)
        return out.reshape((dh, dw) + src.shape[2:])
    xt, yt = _area_tab(w, dw, sx), _area_tab(h, dh, sy)
    # horizontal pass per source row, in table order
    xd = np.array([t[0] for t in xt]); xs = np.array([t[1] for t in xt]); xa = np.array([t[2] for t in xt], np.float32)
    acc = np.zeros((dh, dw, cn), np.float32)
    for dy, sy_, beta in yt:
        buf = np.zeros((dw, cn), np.float32)
        for k in range(len(xt)):                     # entries of one destination column are consecutive: order is preserved
            buf[xd[k]] = buf[xd[k]] + s3[sy_, xs[k]].astype(np.float32) * xa[k]
        acc[dy] = acc[dy] + beta * buf
    return np.clip(cv_round(acc), 0, 255).astype(np.uint8).reshape((dh, dw) + src.shape[2:])


def adjust_black_and_white_point(img: np.ndarray, tpl) -> np.ndarray:
    """image_processors.py:32-41 restated with the same numpy operations."""
    if not tpl:
        return img
    black, white = tpl
    stretched = (np.clip(img, black, white) - black) * (255 / (white - black))
    return stretched.astype(np.uint8)


def seam_voronoi(corners, masks):
    """VoronoiSeamFinder restated with scipy's exact city-block distance transform (independent of the chamfer passes)."""
    from scipy.ndimage import distance_transform_cdt
    masks = [m.copy() for m in masks]
    gap = 10
    n = len(masks)
    for i in range(n - 1):
        for j in range(i + 1, n):
            (x1, y1), (x2, y2) = corners[i], corners[j]
            h1, w1 = masks[i].shape; h2, w2 = masks[j].shape
            rx, ry = max(x1, x2), max(y1, y2)
            rbx, rby = min(x1 + w1, x2 + w2), min(y1 + h1, y2 + h2)
            if not (rx < rbx and ry < rby):
                continue
            rw, rh = rbx - rx, rby - ry

            def cut(m, x0, y0):
                sub = np.zeros((rh + 2 * gap, rw + 2 * gap), np.uint8)
                ys = np.arange(-gap, rh + gap) + ry - y0
                xs = np.arange(-gap, rw + gap) + rx - x0
                vy = (ys >= 0) & (ys < m.shape[0]); vx = (xs >= 0) & (xs < m.shape[1])
                sub[np.ix_(vy, vx)] = m[np.ix_(ys[vy], xs[vx])]
                return sub
            s1, s2 = cut(masks[i], x1, y1), cut(masks[j], x2, y2)
            coll = (s1 != 0) & (s2 != 0)
            u1, u2 = (s1 != 0) & ~coll, (s2 != 0) & ~coll
            big = 65534
            d1 = distance_transform_cdt(~u1, metric="taxicab").astype(np.int64) if u1.any() else np.full(u1.shape, big, np.int64)
            d2 = distance_transform_cdt(~u2, metric="taxicab").astype(np.int64) if u2.any() else np.full(u2.shape, big, np.int64)
            seam = (d1 < d2)[gap:gap + rh, gap:gap + rw]
            v2 = masks[j][ry - y2:ry - y2 + rh, rx - x2:rx - x2 + rw]
            v1 = masks[i][ry - y1:ry - y1 + rh, rx - x1:rx - x1 + rw]
            v2[seam] = 0
            v1[~seam] = 0
    return masks
